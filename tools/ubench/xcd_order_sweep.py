"""Development tool: tile -> XCD orders of the 128x128-family GEMM on the frame's wide shapes, single cold launches (256 MB flush in front of each), min of 9.
variant v = XCD chunks of row-major tiles, 100 + v = 8 row bands, 200 + v = 4 row bands x 2 column halves, 300 + v = 2 x 4.
    python tools/ubench/xcd_order_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toc3d_amd import lib  # noqa: E402

dev = "cuda:0"
flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
S = lambda: torch.cuda.current_stream().cuda_stream


def cold(fn, reps=9):
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


for name, epi, N, K in (("w1|w2", lib.EPI_SWIGLU, 5504, 1024), ("q|k|v", lib.EPI_BIAS, 3072, 1024), ("w3", lib.EPI_RESIDUAL, 1024, 2752)):
    for M in (2898, 3276, 3744, 4096, 6000):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn((N + 127) // 128 * 128, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        if epi == lib.EPI_RESIDUAL:
            out, ldo, res = torch.zeros(M, N, device=dev), N, torch.randn(M, N, device=dev)
        elif epi == lib.EPI_SWIGLU:
            out, ldo, res = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev), N // 2, None
        else:
            out, ldo, res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev), N, None
        row = []
        for base in ((16, 17, 19, 49) if epi != lib.EPI_RESIDUAL else (14, 17, 26)):
            for order in (0, 100, 200, 300):
                v = base + order
                try:
                    t = cold(lambda: lib.call("toc3d_linear_ex", lib.BF16, epi, v, A, K, W, K, b, out, ldo, res, N if res is not None else 0, 0, None, None, M, N, K,
                                              (N // 2 - 22) if epi == lib.EPI_SWIGLU else 0, S()))
                except RuntimeError:
                    t = float("nan")
                row.append(f"v{v}: {t:6.1f}")
        print(f"{name:6s} M={M:5d}  " + "  ".join(row), flush=True)
