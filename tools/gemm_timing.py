#!/usr/bin/env python3
"""Lab: per-phase cycle breakdown of the persistent GEMM (library built with -DVQS_ABLATE=32)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VQS_LIB_PATH"] = os.path.join(ROOT, "build/lab/libvqs_abl32.so")
from t2v_metrics_amd import engine
lib = engine.load_library()
lib.vqs_debug_set_gemm_timing.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(64 * 8 * 8, dtype=torch.int64, device="cuda")
assert lib.vqs_debug_set_gemm_timing(dbg.data_ptr()) == 0
g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, N, K, epi in (("xl_qkv", 155648, 6144, 2048, 0), ("xl_wi", 155648, 10240, 2048, 5), ("xl_o", 155648, 2048, 2048, 0), ("vit_qkv", 147712, 3072, 1024, 0), ("vit_fc1", 147712, 4096, 1024, 1), ("vit_fc2", 147712, 1024, 4096, 0), ("vit_out", 147712, 1024, 1024, 0)):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    resid = torch.randn(M, N, device="cuda", generator=g) if epi == 4 else None
    out = engine.gemm(A, W, epi, resid=resid, variant=7)
    for _ in range(2):
        engine.gemm(A, W, epi, resid=resid, out=out, variant=7)
    torch.cuda.synchronize()
    d = dbg.view(64, 8, 8).double().cpu()
    nk, ntile = d[..., 5].mean().item(), d[..., 6].mean().item()
    print(f"== {tag} M{M} N{N} K{K} epi{epi}: per wave: {ntile:.0f} tiles, {nk:.0f} k-tiles, total {d[...,7].mean().item()/1e3:.0f} kcyc")
    for grp, sl in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
        x = d[:, sl]
        per_kt = [x[..., i].mean().item() / nk for i in range(4)]
        print(f"   {grp}: per k-tile cycles: vmcnt_wait {per_kt[0]:7.0f}  barrier {per_kt[1]:7.0f}  glds_issue {per_kt[2]:7.0f}  compute {per_kt[3]:7.0f}  | sum {sum(per_kt):7.0f};  epilogue/tile {x[...,4].mean().item()/ntile:8.0f}")
