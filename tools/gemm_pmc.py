#!/usr/bin/env python3
"""Run the frame's GEMM shapes with their shipped tile variants (toc3d_amd/tuned/*.json), a few launches each, for
rocprofv3 --pmc passes:   python tools/gemm_pmc.py [table.json] [M,M,...]      (default: the ToC3D_faster table, M = 6000, 3744, 2178)"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toc3d_amd import lib
dev = "cuda:0"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "toc3d_amd", "tuned", "toc3d_faster_320x800_bf16.json")
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6000, 3744, 2178]
for (epi, M, N, K), v in json.load(open(table))["table"]:
    if M not in Ms or K == 768:
        continue
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    if epi == lib.EPI_RESIDUAL:
        out, ldo, res = torch.zeros(M, N, device=dev), N, torch.randn(M, N, device=dev)
    elif epi == lib.EPI_SWIGLU:
        out, ldo, res = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev), N // 2, None
    else:
        out, ldo, res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev), N, None
    for _ in range(5):
        lib.call("toc3d_linear_ex", lib.BF16, epi, v, A, K, W, K, b, out, ldo, res, N if res is not None else 0, 0, None, None, M, N, K,
                 (N // 2 - 22) if epi == lib.EPI_SWIGLU else 0, lib.stream_ptr())
    torch.cuda.synchronize()
    print(f"epi{epi} M={M} N={N} K={K} variant {v}", flush=True)
