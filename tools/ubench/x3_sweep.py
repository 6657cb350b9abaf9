#!/usr/bin/env python3
"""Development: the bf16 x 3 GEMM (f32 operands) on the frame's shapes, every x3 tile variant, warm back-to-back loops; next to the bf16 GEMM's best."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
dev = "cuda:0"
S = lib.stream_ptr
X3V = (1, 8, 10, 14, 16, 17, 19, 22, 26, 28, 49, 116, 117, 122, 126)
BFV = (16, 17, 19, 45, 49, 52, 116, 117, 29, 14, 26)
for name, epi, M, N, K in (("qkv", lib.EPI_BIAS, 6000, 3072, 1024), ("w12", lib.EPI_SWIGLU, 6000, 5504, 1024), ("w3", lib.EPI_RESIDUAL, 6000, 1024, 2752),
                           ("proj", lib.EPI_RESIDUAL, 6000, 1024, 1024), ("w12", lib.EPI_SWIGLU, 3276, 5504, 1024), ("w3", lib.EPI_RESIDUAL, 3276, 1024, 2752)):
    line = f"{name:5s} M={M} N={N} K={K} |"
    for dt, tdt, vs, tag in ((lib.F32X3, torch.float32, X3V, "x3"), (lib.BF16, torch.bfloat16, BFV, "bf16")):
        A = torch.randn(M, K, device=dev).to(tdt)
        W = (torch.randn((N + 127) // 128 * 128, K, device=dev) * K ** -0.5).to(tdt)
        b = torch.randn(N, device=dev)
        ldo = N // 2 if epi == lib.EPI_SWIGLU else N
        out = torch.zeros(M, ldo, dtype=torch.float32 if epi == lib.EPI_RESIDUAL else tdt, device=dev)
        res = torch.randn(M, N, device=dev) if epi == lib.EPI_RESIDUAL else None
        best = {}
        for v in vs:
            def run():
                lib.call("toc3d_linear_ex", dt, epi, v, A, K, W, K, b, out, ldo, res, N if res is not None else 0, 0, None, None, M, N, K, 2730 if epi == lib.EPI_SWIGLU else 0, S())
            try:
                run()
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): run()
                e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 100)
            best[v] = sorted(ts)[2]
        top = sorted(best.items(), key=lambda kv: kv[1])[:3]
        fl = 2.0 * M * N * K
        line += f"  {tag}: " + " ".join(f"v{v}:{t:.0f}us({fl / t / 1e6:.0f}TF)" for v, t in top)
    print(line, flush=True)
