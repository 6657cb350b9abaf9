"""Development yardstick (round 5): the phased 256x256 tile (variants 60 / 160) and the 128x128 family on large shapes, warm back-to-back
loops (20 launches, best of 5), random bf16 operands, next to torch.matmul (hipBLASLt / rocBLAS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from toc3d_amd import lib

S = lib.stream_ptr


def timeit(fn, reps=20, rounds=5):
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


for name, M, N, K in (("square 8192", 8192, 8192, 8192), ("square 4096", 4096, 4096, 4096), ("w1|w2 @ 6x1600x640", 24000, 5504, 1024), ("q|k|v @ 6x1600x640", 24000, 3072, 1024),
                      ("w3 @ 6x1600x640", 24000, 1024, 2752), ("w1|w2 B = 2", 12000, 5504, 1024), ("w1|w2 B = 1", 6000, 5504, 1024)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Wp = torch.zeros((N + 255) // 256 * 256, K, device="cuda", dtype=torch.bfloat16)
    Wp[:N] = W
    t_v = timeit(lambda: torch.matmul(A, W.t(), out=out))
    fl = 2.0 * M * N * K
    row = []
    for v in (16, 116, 17, 117, 149, 60, 160, 61, 62, 63):
        t = timeit(lambda: lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_BIAS, v, A, K, Wp, K, b, out, N, None, 0, 0, None, None, M, N, K, 0, S()), rounds=3)
        row.append(f"v{v}: {t:7.1f} us {fl / t / 1e6:5.0f} TF")
    print(f"{name:20s} M={M:5d} N={N:5d} K={K:5d}: vendor {t_v:7.1f} us {fl / t_v / 1e6:5.0f} TF | " + " | ".join(row), flush=True)
