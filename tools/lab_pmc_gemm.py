#!/usr/bin/env python3
"""PMC subject: one GEMM shape of the path run by GEMM forms of this library (PMC_VARIANTS, default 3 = 8-wave, 10 = quad; PMC_LIB = library) and by
hipBLASLt (torch.matmul), a few launches each -- run under `rocprofv3 --kernel-trace --pmc ...` (tools/gpu_pmc.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (155648, 4096, 4096))]
lib = engine.load_library(os.path.join(ROOT, os.environ.get("PMC_LIB", "t2v_metrics_amd/libvqs_hip.so")))
VARIANTS = [int(x) for x in os.environ.get("PMC_VARIANTS", "3,10").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for v in VARIANTS:
    for _ in range(3):
        assert lib.vqs_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), None, None, M, N, K, K, K, N, 0, 0, 0, v, st) == 0
for _ in range(3):
    torch.matmul(A, W.t(), out=out)
torch.cuda.synchronize()
