"""bench.py's harness on CPU: `python bench.py --gpus 2` must launch its own two ranks (torch.distributed.run on
127.0.0.1), shard the work, all_gather the scores (gloo here, RCCL on the GPU box) and print ONE JSON line whose
result does not depend on the number of ranks.  The HIP engine is replaced by tests/bench_double.py through the
harness self-test hook; nothing here is a measurement."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, gpus):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["VQS_BENCH_ENGINE_DOUBLE"] = "tests.bench_double:DoubleEngine"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--model", "tiny", "--batch", "8", "--cpu-pairs", "0"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # exactly one JSON line (rank 0)
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_gpus_2_self_launches_and_fixed_total_result_is_rank_count_invariant():
    one = _run(["--pairs", "50"], 1)
    two = _run(["--pairs", "50"], 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert len(two["per_rank_pairs_per_s"]) == 2 and all(v > 0 for v in two["per_rank_pairs_per_s"])
    assert one["config"]["total_pairs"] == two["config"]["total_pairs"] == 50
    assert one["scaling"] == two["scaling"] == "strong"
    # same 50 global pairs whatever the sharding: the gathered, input-ordered scores agree
    assert abs(one["scores_checksum"] - two["scores_checksum"]) < 1e-6 * max(1.0, abs(one["scores_checksum"]))
    assert "not a measurement" in two["data"]


@pytest.mark.timeout(900)
def test_weak_scaling_default_and_bucketed_workload_through_the_launcher():
    weak = _run(["--steps", "3", "--warmup", "1"], 2)
    assert weak["scaling"] == "weak" and weak["steps"] == 3 and weak["config"]["total_pairs"] == 2 * 3 * 8
    assert weak["metric"].endswith("tiny") and weak["unit"] == "pairs/s" and weak["value"] > 0
    g1 = _run(["--workload", "genai1600", "--batch", "512"], 1)
    g2 = _run(["--workload", "genai1600", "--batch", "512"], 2)
    assert g1["config"]["total_pairs"] == g2["config"]["total_pairs"] == 9600
    assert abs(g1["scores_checksum"] - g2["scores_checksum"]) < 1e-6 * abs(g1["scores_checksum"])


def test_length_buckets_are_a_permutation_with_minimal_padding():
    sys.path.insert(0, ROOT)
    import bench
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(8, 41, (9600,), generator=g)
    buckets = bench.length_buckets(lens, 256)
    allidx = torch.cat(buckets)
    assert torch.equal(torch.sort(allidx).values, torch.arange(9600))
    waste = sum(int((lens[b].max() - lens[b]).sum()) for b in buckets)
    unsorted_waste = sum(int((lens[s:s + 256].max() - lens[s:s + 256]).sum()) for s in range(0, 9600, 256))
    assert waste <= 9600 and waste * 10 < unsorted_waste       # < 1 padded token per pair vs ~16 unsorted


def test_traffic_provenance_stamps():
    """roofline.traffic is reported only while the PMC record is about the code that runs: the device-code stamp (sha256 of the
    library's .hip_fatbin section) when the record carries one, the source stamp otherwise."""
    import json
    import bench
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "t2v_metrics_amd", "libvqs_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    h = bench.device_code_hash()
    assert h and len(h) == 16 and h == bench.device_code_hash(so)
    assert bench.device_code_hash(__file__) is None                                  # not an ELF file
    ok, how = bench.traffic_stamp_matches({"device_code_sha256_16": h, "csrc_sha256_16": "stale"})
    assert ok and h in how                                                            # the device-code stamp wins
    assert not bench.traffic_stamp_matches({"device_code_sha256_16": "0" * 16, "csrc_sha256_16": bench.csrc_hash()})[0]
    assert bench.traffic_stamp_matches({"csrc_sha256_16": bench.csrc_hash()})[0]    # older records: source stamp
    assert not bench.traffic_stamp_matches({"csrc_sha256_16": "0" * 16})[0]
    rec = json.load(open(os.path.join(os.path.dirname(so), "..", "profiles", "gemm_traffic_xxl_b256.json")))
    assert "device_code_sha256_16" in rec and "traffic_bytes_per_launch" in rec
    # the finest stamp: machine code + descriptors of the GEMM kernels alone -- what the counters were collected on.  The committed
    # record must be about the kernels this tree builds (an edit of a GEMM kernel without a new PMC run fails HERE, not silently on the box)
    g = bench.gemm_kernels_hash()
    assert g and len(g) == 16 and g == bench.gemm_kernels_hash(so) and bench.gemm_kernels_hash(__file__) is None
    ks = bench.device_kernels(so)
    assert len(ks) > 100 and all(len(kd) == 64 and len(code) > 0 for code, kd in ks.values())
    assert sum("gemm_bf16_quad" in n for n in ks) == 5 and sum("gemm_f16_quad" in n for n in ks) == 4 and sum("gemm_f16b_quad" in n for n in ks) == 2
    ok, how = bench.traffic_stamp_matches({"gemm_kernels_sha256_16": g, "device_code_sha256_16": "stale"})
    assert ok and g in how                                                            # ... and it wins over the whole-library stamp
    assert not bench.traffic_stamp_matches({"gemm_kernels_sha256_16": "0" * 16, "device_code_sha256_16": h})[0]
    # the committed record names the kernel families it was collected on (round 5: bf16, fp16 and fp16-operand / bf16-result quad kernels)
    ok, how = bench.traffic_stamp_matches(rec)
    assert ok, "profiles/gemm_traffic_xxl_b256.json was measured on other GEMM kernels (%s): re-run tools/gpu_pmc_bench.sh" % how


def test_pmc_summary_counts_both_gemm_kernel_families_and_stamps_the_record(tmp_path):
    """tools/pmc_summary.py on a synthetic rocprofv3 database (two passes: FETCH_SIZE, WRITE_SIZE): the per-launch traffic averages over
    the bf16 AND the fp16 GEMM launches, other kernels are left out, the record says whether fp16 launches were in the collection and carries
    the stamps bench.traffic_stamp_matches checks -- so that the next collection on the GPU box does not fail on the summary step."""
    import json
    import sqlite3
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "t2v_metrics_amd", "libvqs_hip.so")):
        pytest.skip("library not built")
    rows = {"FETCH_SIZE": [("void vqs::gemm_bf16_quad<5>(vqs::GemmParams)", 65536, 3, 4000.0), ("void vqs::gemm_f16b_quad<0>(vqs::GemmParams)", 65536, 1, 2000.0),
                           ("void vqs::attn_fwd_dma_kernel<true>(vqs::AttnParams)", 1024, 5, 9e9)],
            "WRITE_SIZE": [("void vqs::gemm_bf16_quad<5>(vqs::GemmParams)", 65536, 3, 1000.0), ("void vqs::gemm_f16_quad<0>(vqs::GemmParams)", 65536, 1, 3000.0)]}
    for with_f16 in (True, False):
        d = tmp_path / ("f16" if with_f16 else "bf16")
        d.mkdir()
        for i, (ctr, rs) in enumerate(rows.items()):
            con = sqlite3.connect(str(d / f"pass{i + 1}_results.db"))
            con.execute("create table counters_collection (kernel_name text, counter_name text, grid_size int, value real, start int, end int)")
            for kn, gs, n, v in rs:
                if "gemm_f16" in kn and not with_f16:
                    continue
                for _ in range(n):
                    con.execute("insert into counters_collection values (?,?,?,?,?,?)", (kn, ctr, gs, v, 0, 1000))
            con.commit()
            con.close()
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(d), "vqs::"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-800:]
        rec = json.load(open(d / "gemm_traffic.json"))
        n = 4 if with_f16 else 3
        fetch = 2 * 1024 * ((3 * 4000.0 + (2000.0 if with_f16 else 0.0)) / n)       # KiB -> bytes, x2: gfx950 tallies 128-B requests at 64 B
        write = 1024 * ((3 * 1000.0 + (3000.0 if with_f16 else 0.0)) / n)
        assert rec["launches_per_pass"] == n and rec["vit_fp16"] is with_f16
        assert rec["fetch_bytes_per_launch"] == pytest.approx(fetch) and rec["write_bytes_per_launch"] == pytest.approx(write)
        assert rec["traffic_bytes_per_launch"] == pytest.approx(fetch + write)
        assert rec["gemm_kernels_patterns"] == ["gemm_bf16_", "gemm_f16"]            # "gemm_f16": gemm_f16_quad (fp16 result) and gemm_f16b_quad (bf16 result)
        assert rec["gemm_kernels_sha256_16"] == bench.gemm_kernels_hash(patterns=("gemm_bf16_", "gemm_f16")) != bench.gemm_kernels_hash()
        ok, how = bench.traffic_stamp_matches(rec)
        assert ok and rec["gemm_kernels_sha256_16"] in how
