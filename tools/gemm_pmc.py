#!/usr/bin/env python3
"""Run one GEMM shape / variant repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toc3d_amd import lib
dev = "cuda:0"
M, N, K, epi = 6192, 3072, 1024, lib.EPI_BIAS
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8, 13]
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
for v in variants:
    for _ in range(5):
        lib.call("toc3d_linear_ex", lib.BF16, epi, v, A, K, W, K, b, out, N, None, 0, 0, None, 0, M, N, K, 0, lib.stream_ptr())
torch.cuda.synchronize()
# residual-epilogue shape (w3)
M, N, K = 6192, 1024, 2752
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
res = torch.randn(M, N, device=dev)
o32 = torch.zeros(M, N, device=dev)
for v in variants:
    for _ in range(5):
        lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_RESIDUAL, v, A, K, W, K, b[:N], o32, N, res, N, 0, None, 0, M, N, K, 0, lib.stream_ptr())
torch.cuda.synchronize()
