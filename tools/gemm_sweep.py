#!/usr/bin/env python3
"""GPU micro-benchmark: every toc3d_linear_ex tile/pipeline variant on the GEMM shapes of the ToC3D_faster frame.
Prints TFLOP/s per (shape, epilogue, variant); interleaved rounds (within-process A/B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toc3d_amd import lib

dev = "cuda:0"
dt, tdt = lib.BF16, torch.bfloat16
if len(sys.argv) > 1 and sys.argv[1] == "fp32":
    dt, tdt = lib.F32, torch.float32
S = lib.stream_ptr
C, Hp = 1024, 2752
shapes = []   # (name, epi, M, N, K)
for M in (6000, 3744):
    shapes += [("qkv", lib.EPI_BIAS, M, 3072, 1024), ("proj", lib.EPI_RESIDUAL, M, 1024, 1024),
               ("w3", lib.EPI_RESIDUAL, M, 1024, Hp)] + ([("w12", lib.EPI_SWIGLU, M, 2 * Hp, 1024)] if not os.environ.get("NO_SWIGLU") else [])
variants = [int(v) for v in os.environ.get('VARIANTS', '16,17,19,38,39,40,41,42').split(',')]
res = {}
for name, epi, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(tdt)
    W = (torch.randn((N + 127) // 128 * 128, K, device=dev) * K ** -0.5).to(tdt)
    b = torch.randn(N, device=dev)
    ldo = N // 2 if epi == lib.EPI_SWIGLU else N
    out = torch.zeros(M, ldo, dtype=torch.float32 if epi == lib.EPI_RESIDUAL else tdt, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == lib.EPI_RESIDUAL else None
    ref = None
    for v in variants:
        def run():
            lib.call("toc3d_linear_ex", dt, epi, v, A, K, W, K, b, out, ldo, resid, N if resid is not None else 0, 0, None, 0, M, N, K, 2730 if epi == lib.EPI_SWIGLU else 0, S())
        run(); torch.cuda.synchronize()
        cur = out.float().clone()
        if ref is None: ref = cur
        else: assert (cur - ref).abs().max().item() <= 1e-2 * ref.abs().max().item(), (name, v)
    times = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.call("toc3d_linear_ex", dt, epi, v, A, K, W, K, b, out, ldo, resid, N if resid is not None else 0, 0, None, 0, M, N, K, 2730 if epi == lib.EPI_SWIGLU else 0, S())
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    line = f"{name:5s} M={M:5d} N={N:5d} K={K:5d} | " + " ".join(f"v{v}:{fl / (sorted(times[v])[2] * 1e-3) / 1e12:6.0f}TF({sorted(times[v])[2]*1e3:5.1f}us)" for v in variants)
    print(line, flush=True)
