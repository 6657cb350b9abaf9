#!/usr/bin/env python3
"""Development: what the folded-LayerNorm epilogues cost their GEMMs -- w1|w2 as EPI_SWIGLU_STATS (epi4) vs EPI_SWIGLU_STATS_LN (epi7), proj as EPI_RESIDUAL (1) vs
EPI_RESIDUAL_STATS (6), w3 as EPI_RESIDUAL (1) vs EPI_RESIDUAL_LN (5); same tile, warm back-to-back loops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
dev = "cuda:0"; S = lib.stream_ptr
C, Hd, Hp = 1024, 2730, 2752
def timeit(fn):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    return sorted(ts)[3]
for M in (6000, 3276):
    a = torch.randn(M, C, device=dev).bfloat16(); w12 = (torch.randn(2 * Hp, C, device=dev) * C ** -0.5).bfloat16(); b12 = torch.randn(2 * Hp, device=dev)
    c1 = torch.randn(2 * Hp, device=dev); hid = torch.zeros(M, Hp, dtype=torch.bfloat16, device=dev)
    cap, cap2 = 44, 16
    st = torch.zeros(4 + M * cap * 2, device=dev); st2 = torch.rand(4 + M * cap2 * 2, device=dev) + 1.0
    for v in (16, 116, 17):
        t4 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 4, v, a, C, w12, C, b12, hid, Hp, None, 0, 0, None, None, M, 2 * Hp, C, Hd, st, cap, None, 0, None, 0, 0.0, None, 0, None, S()))
        t7 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 7, v, a, C, w12, C, b12, hid, Hp, None, 0, 0, None, None, M, 2 * Hp, C, Hd, st, cap, st2, cap2 | 16 << 32, c1, C, 1e-6, None, 0, None, S()))
        t2 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 2, v, a, C, w12, C, b12, hid, Hp, None, 0, 0, None, None, M, 2 * Hp, C, Hd, *lib.NO_FUSED, S()))
        print(f"w12 M={M} v{v}: SWIGLU {t2:.1f}  SWIGLU_STATS {t4:.1f}  SWIGLU_STATS_LN {t7:.1f} us", flush=True)
    att = torch.randn(M, C, device=dev).bfloat16(); wp = (torch.randn(C, C, device=dev) * C ** -0.5).bfloat16(); bp = torch.randn(C, device=dev)
    x = torch.randn(M, C, device=dev); acopy = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    for v in (17, 126, 26):
        t1 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 1, v, att, C, wp, C, bp, x, C, x, C, 0, None, None, M, C, C, 0, *lib.NO_FUSED, S()))
        t6 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 6, v, att, C, wp, C, bp, x, C, x, C, 0, None, None, M, C, C, 0, st2, cap2, None, 0, None, 0, 0.0, acopy, C, None, S()))
        print(f"proj M={M} v{v}: RESIDUAL {t1:.1f}  RESIDUAL_STATS {t6:.1f} us", flush=True)
    h = torch.randn(M, Hp, device=dev).bfloat16(); w3 = (torch.randn(C, Hp, device=dev) * Hp ** -0.5).bfloat16(); c13 = torch.randn(C, device=dev)
    for v in (117, 29, 55):
        t1 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 1, v, h, Hp, w3, Hp, bp, x, C, x, C, 0, None, None, M, C, Hp, 0, *lib.NO_FUSED, S()))
        t5 = timeit(lambda: lib.call("toc3d_linear_fused", lib.BF16, 5, v, h, Hp, w3, Hp, bp, x, C, x, C, 0, None, None, M, C, Hp, 0, None, 0, st, cap | 43 << 32, c13, Hd, 1e-6, None, 0, None, S()))
        print(f"w3 M={M} v{v}: RESIDUAL {t1:.1f}  RESIDUAL_LN {t5:.1f} us", flush=True)
