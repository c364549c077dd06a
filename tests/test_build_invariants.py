"""Build-time invariants of the gfx950 code objects, checked from the compiler's own kernel metadata (no GPU needed):
no kernel spills or uses scratch, every kernel fits the CU's 160 KiB of LDS, and the attention kernels keep the LDS
budget their occupancy depends on.  (A 512-byte static array once took the T5 attention kernel from three to two
workgroups per CU and cost 24 % of its rate -- profiles/r1_call90_attn_static_lds_regression.md.)

The device-only assembly of each source is cached under build/isa/ keyed by the sources' size and mtime, so only a
changed file is recompiled (about 20 s for gemm.hip)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "t2v_metrics_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
LDS_PER_CU = 160 * 1024
LDS_GRANULE = 1280            # gfx950 hands out LDS in 1 280-byte granules (128 per CU)
EXTRA_FLAGS = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]}   # as in csrc/Makefile

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc (cross-compiles without a GPU)")


def _kernels(src):
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "vqs_kernels.h")]
    key = "_".join("%d-%d" % (os.path.getsize(d), int(os.path.getmtime(d))) for d in deps)
    out_dir = os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    asm = os.path.join(out_dir, "%s.%s.s" % (src, key))
    if not os.path.exists(asm):
        for old in os.listdir(out_dir):
            if old.startswith(src + "."):
                os.remove(os.path.join(out_dir, old))
        hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *EXTRA_FLAGS.get(src, []), "-S",
                               "--cuda-device-only", "-o", asm, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    text = open(asm).read()
    meta = text[text.index("amdhsa.kernels"):]
    out = {}
    for entry in meta.split("  - .agpr_count:")[1:]:
        def field(name):
            return int(re.search(r"\.%s:\s+(\d+)" % name, entry).group(1))
        out[re.search(r"\.name:\s+(\S+)", entry).group(1)] = {
            "lds": field("group_segment_fixed_size"), "scratch": field("private_segment_fixed_size"),
            "vgpr": field("vgpr_count"), "vgpr_spill": field("vgpr_spill_count"), "wg": field("max_flat_workgroup_size"),
            "agpr": int(entry.split()[0])}
    assert out, "no kernel metadata found in " + asm
    return out


@pytest.mark.parametrize("src", ["gemm.hip", "attn.hip", "elementwise.hip"])
def test_no_kernel_spills_or_exceeds_the_cu(src):
    for name, k in _kernels(src).items():
        assert k["vgpr_spill"] == 0 and k["scratch"] == 0, (name, k)
        assert k["lds"] <= LDS_PER_CU, (name, k)
        # a 512-thread workgroup is two waves per SIMD: at most 256 registers per lane each
        assert k["vgpr"] <= (256 if k["wg"] > 256 else 512), (name, k)


def _resident(static_lds, dynamic_lds):
    per_wg = -(-(static_lds + dynamic_lds) // LDS_GRANULE) * LDS_GRANULE
    return LDS_PER_CU // per_wg


def test_attention_lds_budget_keeps_its_occupancy():
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    ks = _kernels("attn.hip")
    dma = {n: k for n, k in ks.items() if "attn_fwd_dma_kernel" in n}
    hd = {n: k for n, k in ks.items() if "attn_fwd_hd_kernel" in n}
    assert len(dma) == 2 and len(hd) == 2
    for name, k in {**dma, **hd}.items():
        assert k["lds"] == 0, (name, "static LDS in an attention kernel changes its occupancy", k)
    # T5-XL / XXL encoder (S = 32-token prompt + 576 patches = 608, position bias): three workgroups per CU
    t5 = lib.vqs_attention_lds_bytes(608, 1, 0)
    assert t5 == 2 * 16384 + 324 * 64 and _resident(0, t5) == 3
    # CLIP ViT-L/14-336 (S = 577, no bias): four
    assert lib.vqs_attention_lds_bytes(577, 0, 0) == 2 * 16384 and _resident(0, 2 * 16384) == 4
    # Qwen2.5-VL head-128 kernel: two stages of 64-key K and V tiles with 256-B rows, two workgroups
    assert lib.vqs_attention_lds_bytes(808, 0, 128) == 65536 and _resident(0, 65536) == 2
    assert lib.vqs_attention_lds_bytes(0, 0, 0) == -1 and lib.vqs_attention_lds_bytes(64, 0, 96) == -1


def test_gemm_kernels_own_the_cu():
    """One GEMM workgroup per CU by construction: two 64-KiB stages (+ the 2-KiB touch sink / row-reduction scratch of the
    variants that have one) leave no room for a second, and the persistent grid is sized for that.  The 8-wave forms are
    512 threads (two waves per SIMD, <= 256 registers per lane); the wide form is 256 threads -- ONE wave per SIMD, whose 256
    fp32 accumulators per lane are the whole AGPR file (hand-allocated, gemm.hip mfma_fixed) next to <= 256 VGPRs."""
    wide = 0
    for name, k in _kernels("gemm.hip").items():
        if "gemm_bf16" in name:
            assert 2 * 65536 <= k["lds"] <= 2 * 65536 + 8192, (name, k)
            assert _resident(k["lds"], 0) == 1, (name, k)
            if "gemm_bf16_wide" in name:
                wide += 1
                assert k["wg"] == 256 and k["agpr"] == 256 and k["vgpr"] <= 512, (name, k)
            else:
                assert k["wg"] == 512 or "gemm_bf16_ws" in name, (name, k)
    assert wide == 5, "wide form: epilogues bf16, quick_gelu, erf-GELU, gated, head-major"


def test_shipped_library_has_no_lab_code_and_reads_no_environment():
    """Hygiene of libvqs_hip.so: the A/B forms kept for the record (wave-specialised GEMM, register-staged GEMM and
    attention) are compiled only under -DVQS_LAB (make lab -> build/lab/, never loaded by the package), and the product
    library does not import getenv: execution forms are chosen through vqs_set_option, not through the environment."""
    names = set(_kernels("gemm.hip")) | set(_kernels("attn.hip"))
    assert not [n for n in names if "gemm_bf16_ws" in n or "attn_fwd_kernel" in n], names
    assert not [n for n in names if re.search(r"gemm_bf16_kernelILi\d+ELi1EE", n)], "register-staged GEMM (variant 1) instantiated"
    assert any("gemm_bf16_persistent" in n for n in names) and any("attn_fwd_dma_kernel" in n for n in names)
    lib = os.path.join(ROOT, "t2v_metrics_amd", "libvqs_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib], text=True)
    assert "getenv" not in undefined, "the shipped library must not read environment variables"
    for src in ("gemm.hip", "attn.hip", "elementwise.hip", "vqs_api.cpp", "vqs_qwen.cpp"):
        text = open(os.path.join(CSRC, src)).read()
        outside = re.sub(r"#ifdef VQS_LAB.*?#endif", "", text, flags=re.S)
        assert "getenv" not in outside, src + ": getenv outside an #ifdef VQS_LAB block"
        assert "VQS_ABLATE" not in text and "VQS_ATTN_ABLATE" not in text, src + ": ablation scaffolding is back"
