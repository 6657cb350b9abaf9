"""Development yardstick (not used by the product): what does the vendor GEMM behind torch.matmul (hipBLASLt / rocBLAS) reach on
the frame's shapes, next to toc3d_linear_ex?  Plain bias-free bf16 GEMMs, back-to-back loops of 20 launches, best of 5."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
S = lib.stream_ptr
def timeit(fn, reps=20, rounds=5):
    best = 1e9
    for _ in range(rounds):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3
for name, M, N, K in (("qkv dense", 6000, 3072, 1024), ("w12 dense", 6000, 5504, 1024), ("w3 dense", 6000, 1024, 2752), ("proj dense", 6000, 1024, 1024),
                      ("qkv accel", 3744, 3072, 1024), ("w12 accel", 3744, 5504, 1024), ("w3 accel", 2808, 1024, 2752), ("square 8192", 8192, 8192, 8192)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16(); b = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Wp = torch.zeros((N + 127) // 128 * 128, K, device="cuda", dtype=torch.bfloat16); Wp[:N] = W
    t_v = timeit(lambda: torch.matmul(A, W.t(), out=out))
    best = None
    for v in (16, 17, 8, 15, 26, 29, 116, 117, 126):
        t = timeit(lambda: lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_BIAS, v, A, K, Wp, K, b, out, N, None, 0, 0, None, None, M, N, K, 0, S()), rounds=3)
        best = (t, v) if best is None or t < best[0] else best
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:5d} N={N:5d} K={K:5d}: vendor {t_v:7.1f} us = {fl / t_v / 1e6:6.0f} TF | toc3d_linear_ex v{best[1]:<3d} {best[0]:7.1f} us = {fl / best[0] / 1e6:6.0f} TF", flush=True)
