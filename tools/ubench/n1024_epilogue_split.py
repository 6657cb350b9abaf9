#!/usr/bin/env python3
"""Development measurement (round 5; VERDICT r04 weak 2: "nothing in profiles/ separates the two causes per launch"): the N = 1024 residual GEMMs (attn.proj K = 1024, mlp.w3
K = 2752) at the frame's row counts, the SAME tile, three epilogues that differ only in the bytes they move per output element:
    bias              bf16 out                                    2 B
    residual          f32 residual in, f32 out (in place)         8 B
    residual_stats    + bf16 copy + (sum, sum^2) per 64 columns  10 B + statistics
warm = back-to-back loop of 20 (best of 5), cold = single launches behind a 512 MB memset (lower quartile of 9).  The bias row is the K loop + launch cost of the tile; the
differences are what the epilogue's memory traffic costs on top."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from toc3d_amd import lib

S = lib.stream_ptr
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def warm(fn, reps=20, rounds=5):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


def cold(fn, n=9):
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 4]


C = 1024
print("# shape | tile | bias (2 B/elem) warm / cold us | residual (8 B) | residual_stats (10 B + statistics) | algorithmic bytes of the three epilogues (MB) | K-loop operand bytes (MB)")
for M, K, v in ((6000, 1024, 17), (3744, 1024, 26), (3276, 1024, 126), (2808, 1024, 14), (6000, 2752, 155), (3744, 2752, 29), (3276, 2752, 14), (2808, 2752, 114)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(C, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(C, device="cuda")
    ob = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, C, device="cuda")
    araw = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    cap = C // 64
    st = torch.zeros(4 + M * cap * 2, device="cuda")
    f_bias = lambda: lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_BIAS, v, A, K, W, K, b, ob, C, None, 0, 0, None, None, M, C, K, 0, S())
    f_res = lambda: lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_RESIDUAL, v, A, K, W, K, b, x, C, x, C, 0, None, None, M, C, K, 0, S())
    f_stats = lambda: lib.call("toc3d_linear_fused", lib.BF16, lib.EPI_RESIDUAL_STATS, v, A, K, W, K, b, x, C, x, C, 0, None, None, M, C, K, 0,
                               st, cap, None, 0, None, 0, 0.0, araw, C, None, S())
    row = [f"{warm(f):6.1f} / {cold(f):6.1f}" for f in (f_bias, f_res, f_stats)]
    mb = [M * C * e / 1e6 for e in (2, 8, 10)]
    print(f"M={M:5d} N=1024 K={K:4d} | v{v:<3d} | " + " | ".join(row) + f" | {mb[0]:5.1f} {mb[1]:5.1f} {mb[2]:5.1f} | {(M + C) * K * 2 / 1e6:5.1f}", flush=True)
