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
    incs = {"gemm.hip": ["gemm_quad.inc", "gemm_quad_kernel.inc", "gemm_stream.inc"], "attn.hip": ["attn_dma_kernel.inc"]}.get(src, [])
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "vqs_kernels.h")] + [os.path.join(CSRC, i) for i in incs]
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
    out["__asm__"] = text
    return out


@pytest.mark.parametrize("src", ["gemm.hip", "attn.hip", "elementwise.hip", "qwen_decode.hip"])
def test_no_kernel_spills_or_exceeds_the_cu(src):
    for name, k in _kernels(src).items():
        if name == "__asm__":
            continue
        assert k["vgpr_spill"] == 0 and k["scratch"] == 0, (name, k)
        assert k["lds"] <= LDS_PER_CU, (name, k)
        # a 512-thread workgroup is two waves per SIMD: at most 256 registers per lane each
        assert k["vgpr"] <= (256 if k["wg"] > 256 else 512), (name, k)


@pytest.mark.parametrize("src", ["gemm.hip", "attn.hip", "elementwise.hip", "qwen_decode.hip"])
def test_no_valu_written_sgpr_feeds_an_inline_asm_memory_instruction_within_five_wait_states(src):
    """gfx9 hazard: an SGPR written by a VALU instruction (v_readlane_b32 = the reload of a spilled scalar, v_readfirstlane_b32) must not be read
    by a vector-memory instruction -- as descriptor or scalar offset, or through M0 by an LDS-DMA -- within the next 5 wait states.  The compiler
    pads its own instructions, but its hazard recogniser does not look into INLINE ASM, which is how this library issues its scheduled LDS-DMA
    loads: a lab kernel of round 5 stored one of 32 row pieces to the wrong row exactly this way (profiles/r5_gemm_gap.md section 8).  Checked
    on the ISA of every kernel: no VALU-written SGPR is an operand of a memory instruction (or M0 write) INSIDE an asm statement fewer than 5
    instructions later (each instruction is at least one wait state; s_nop N counts N + 1; a scalar instruction re-defining the register ends
    the window -- its consumers read the scalar unit's value)."""
    text = _kernels(src)["__asm__"]
    found, n_asm_mem, n_hint = [], 0, 0
    for m in re.finditer(r"^(_ZN3vqs\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M):
        code, in_asm = [], False
        for raw in m.group(2).splitlines():
            if "#ASMSTART" in raw:
                in_asm = True
            elif "#ASMEND" in raw:
                in_asm = False
            ln = raw.split(";")[0].strip()
            if ln and not ln.startswith(".") and not ln.endswith(":"):
                code.append((ln, in_asm))
        n_asm_mem += sum(1 for ln, a in code if a and ln.startswith(("buffer_", "global_", "flat_")))
        for i, (ln, _) in enumerate(code):
            if not ln.startswith(("v_readlane_b32", "v_readfirstlane_b32")):
                continue
            dst = ln.split()[1].rstrip(",")
            if not re.fullmatch(r"s\d+", dst):
                continue
            n, waited = int(dst[1:]), 0
            for nxt, in_asm in code[i + 1:i + 6]:
                if waited >= 5:
                    break
                if nxt.startswith("s_") and re.match(r"s_\w+ %s\b" % dst, nxt):
                    break
                if in_asm and nxt.startswith(("buffer_", "global_", "flat_")):
                    regs = set(int(x) for x in re.findall(r"\bs(\d+)\b", nxt))
                    for a, b in re.findall(r"s\[(\d+):(\d+)\]", nxt):
                        regs |= set(range(int(a), int(b) + 1))
                    # tolerated: the persistent kernels' L2 TOUCH (`buffer_load_dword ... lds`: one dword per lane into a sink nobody reads, a
                    # prefetch hint whose asm block opens with three scalar instructions) -- a stale scalar offset there fetches another line
                    hint = nxt.startswith("buffer_load_dword ") and nxt.endswith("lds")
                    if n in regs and not hint:
                        found.append((m.group(1), ln, nxt))
                    n_hint += 1 if (n in regs and hint) else 0
                if in_asm and nxt.startswith("s_mov_b32 m0") and re.search(r"\b%s\b" % dst, nxt):
                    found.append((m.group(1), ln, nxt))
                waited += (int(nxt.split()[1]) + 1) if nxt.startswith("s_nop") else 1
    assert not found, found[:5]
    if src in ("gemm.hip", "attn.hip"):
        assert n_asm_mem >= 32, n_asm_mem          # the scan saw the asm statements it is about (ASMSTART / ASMEND markers present)
    assert n_hint <= 40, n_hint                    # (the tolerated hint loads: 36 at the time of writing, TOUCH instantiations only)


def test_decode_attention_fits_its_largest_cache_in_lds():
    """qwen_decode_attn_kernel keeps one fp32 score per cached position in dynamic LDS next to its static q / partial-row buffers:
    the launcher's Lmax limit (36 864 positions) must fit the CU together with them."""
    ks = _kernels("qwen_decode.hip")
    k = [v for n, v in ks.items() if "qwen_decode_attn_kernel" in n]
    assert len(k) == 3                                  # <PRECISE = false> (the decode step), <true> (the precise tail over bf16 K / V) and <true, KVH> (over the fp16 forms' K / V)
    for v in k:
        assert v["lds"] + 36864 * 4 <= LDS_PER_CU, v
    src = open(os.path.join(CSRC, "qwen_decode.hip")).read()
    assert "Lmax > 36864" in src


def _resident(static_lds, dynamic_lds):
    per_wg = -(-(static_lds + dynamic_lds) // LDS_GRANULE) * LDS_GRANULE
    return LDS_PER_CU // per_wg


def test_attention_lds_budget_keeps_its_occupancy():
    from t2v_metrics_amd import engine
    lib = engine.load_library()
    ks = _kernels("attn.hip")
    ks = {n: k for n, k in ks.items() if n != "__asm__"}
    dma = {n: k for n, k in ks.items() if "attn_fwd_dma_kernel" in n}
    hd = {n: k for n, k in ks.items() if "attn_fwd_hd_kernel" in n}
    assert len(dma) == 2 and len(hd) == 4              # hd: <causal> x <bf16 | fp16 tensors (the Qwen row's range-safe fp16 forms)>
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
    variants that have one; the quad form's 32 KiB of epilogue scratch went with round 6's register-direct epilogue) leave no room for a second, and the
    persistent grid is sized for that.  The 8-wave forms are 512 threads (two waves per SIMD, <= 256 registers per lane); the
    quad form is 256 threads -- ONE wave per SIMD, whose 256 fp32 accumulators per lane are the whole AGPR file."""
    quad = stream = 0
    ks = _kernels("gemm.hip")
    for name, k in ks.items():
        if name == "__asm__" or "gemm_bf16" not in name:
            continue
        assert _resident(k["lds"], 0) == 1, (name, k)
        if "gemm_bf16_quad" in name:
            quad += 1
            assert k["lds"] == 2 * 65536 and k["wg"] == 256 and k["vgpr"] <= 512, (name, k)      # two K-tile buffers; the register-direct epilogue (round 6) needs no scratch
        elif "gemm_bf16_stream" in name:
            # stream form (round 4): five 32-KiB slab stages = all 160 KiB, four waves, no scratch
            stream += 1
            assert k["lds"] == 5 * 32768 and k["wg"] == 256 and k["scratch"] == 0, (name, k)
        else:
            assert 2 * 65536 <= k["lds"] <= 2 * 65536 + 8192, (name, k)
            assert k["wg"] == 512, (name, k)
    assert quad == 5, "quad form: epilogues bf16, quick_gelu, erf-GELU, gated, head-major"
    assert stream == 6, "stream form: fp32 and bf16 (+ split) results x operands bf16 / fp16 (fp16 result) / fp16 (bf16 result)"


def test_quad_form_keeps_the_compiler_out_of_its_accumulators():
    """gemm_bf16_quad names its 256 accumulator registers itself (a[4 Q : 4 Q + 3] per 16 x 16 block, gemm_quad.inc) in asm
    statements the register allocator cannot see into.  That is sound only while the compiler puts nothing of its own
    into AGPRs -- no fragment, no spill slot, no copy (one draft of the form had `ds_read_b128 a[0:3], ...` in a cold path:
    silent corruption of an accumulator).  Checked on the ISA of every instantiation: the kernel descriptor reserves exactly
    256 AGPRs behind the VGPRs, an AGPR appears only as the C/D operand of a v_mfma or the source of a v_accvgpr_read, there is
    no v_accvgpr_write / scratch access, and the K loop's instruction mix is the scheduled one."""
    text = _kernels("gemm.hip")["__asm__"]
    found = 0
    for m in re.finditer(r"^(_ZN3vqs14gemm_bf16_quadILi(\d+)EEEvNS_10GemmParamsE):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M):
        found += 1
        name, body = m.group(1), m.group(3)
        desc = text[text.index(".amdhsa_kernel " + name):]
        desc = desc[:desc.index(".end_amdhsa_kernel")]
        nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1))
        assert nxt - acc == 256 and acc <= 256, (name, nxt, acc)
        code = [l.split(";")[0].strip() for l in body.splitlines()]
        code = [l for l in code if l and not l.startswith((".", ";")) and not l.endswith(":")]
        agpr = re.compile(r"\ba(\[\d+(:\d+)?\]|\d+)\b")
        bad = [l for l in code if agpr.search(l) and not l.startswith(("v_mfma_f32_16x16x32_bf16", "v_accvgpr_read_b32"))]
        assert not bad, (name, bad[:5])
        assert not [l for l in code if l.startswith(("v_accvgpr_write", "scratch_", "v_accvgpr_mov"))], name
        for l in code:
            if l.startswith("v_mfma"):
                ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
                assert ops[0].startswith("a[") and ops[1].startswith("v[") and ops[2].startswith("v[") and (ops[3] == "0" or ops[3] == ops[0]), l
        n_mfma = sum(l.startswith("v_mfma") for l in code)
        assert n_mfma == 3 * 128, (name, n_mfma)                    # FIRST / MID / LAST K-tile bodies, nothing unrolled twice
        assert sum(" lds" in l and l.startswith("buffer_load_dwordx4") for l in code) == 3 * 16 + 2 * 16, name   # + the two K-tiles of the prologue
    assert found == 5


def test_shipped_library_has_no_lab_code_and_reads_no_environment():
    """Hygiene of libvqs_hip.so: the A/B forms of rounds 1-3 that lost every measurement (wave-specialised, ring, wide and
    register-staged GEMMs, register-staged attention) are gone from the tree (git history keeps them; round 4 removed the
    -DVQS_LAB flavour altogether), and the product library does not import getenv: execution forms are chosen through
    vqs_set_option, not through the environment."""
    names = (set(_kernels("gemm.hip")) | set(_kernels("attn.hip"))) - {"__asm__"}
    assert not [n for n in names if "gemm_bf16_ws" in n or "attn_fwd_kernel" in n or "gemm_bf16_wide" in n or "gemm_bf16_ring" in n], names
    assert not [n for n in names if re.search(r"gemm_bf16_kernelILi\d+ELi1EE", n)], "register-staged GEMM (variant 1) instantiated"
    assert any("gemm_bf16_persistent" in n for n in names) and any("attn_fwd_dma_kernel" in n for n in names)
    lib = os.path.join(ROOT, "t2v_metrics_amd", "libvqs_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib], text=True)
    assert "getenv" not in undefined, "the shipped library must not read environment variables"
    for src in ("gemm.hip", "attn.hip", "elementwise.hip", "vqs_api.cpp", "vqs_qwen.cpp"):
        text = open(os.path.join(CSRC, src)).read()
        assert "getenv" not in text and "VQS_LAB" not in text, src + ": environment switch / lab flavour is back"
        assert "VQS_ABLATE" not in text and "VQS_ATTN_ABLATE" not in text, src + ": ablation scaffolding is back"
    assert not os.path.exists(os.path.join(CSRC, "lab")), "csrc/lab is back"


def _hipcc_version():
    try:
        out = subprocess.check_output([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], text=True, stderr=subprocess.STDOUT)
        return " / ".join(l.strip() for l in out.splitlines() if l.startswith(("HIP version", "AMD clang version")))
    except Exception:                                       # noqa: BLE001
        return "unknown"


def test_the_tree_builds_to_the_device_code_the_gpu_records_were_made_with():
    """profiles/validated_device_code.json names the device code (sha256 of the library's .hip_fatbin section; the build is deterministic)
    that the last GPU runs of the suite / the bench were made with.  A device-code edit makes this fail on the CPU, before anything is
    claimed about it: re-run the GPU suite, then update the record with the new hash and the new logs."""
    import json
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "validated_device_code.json")))
    so = os.path.join(ROOT, "t2v_metrics_amd", "libvqs_hip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    # the hashes pin the TREE only on the toolchain that made the record: another hipcc builds other bytes from the same source
    here = _hipcc_version()
    if rec.get("hipcc_version") and here != rec["hipcc_version"]:
        pytest.skip("record made with %r, this box has %r: the hash comparison would test the toolchain, not the tree" % (rec["hipcc_version"], here))
    assert bench.device_code_hash(so) == rec["device_code_sha256_16"], \
        "the shipped device code differs from the GPU-validated build: run the -m gpu suite and update profiles/validated_device_code.json"
    assert bench.gemm_kernels_hash(so) == rec["gemm_kernels_sha256_16"]
    for ref in rec["validated_by"]:
        assert os.path.exists(os.path.join(ROOT, ref.split(" ")[0])), ref
