"""Development sweep: every tile variant on the N=1024 residual GEMM shapes, single launches (event timed)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
S_ = lib.stream_ptr
dt, tdt = lib.BF16, torch.bfloat16
variants = list(range(1, 43)) + [110, 114, 116, 117, 126]
for (M, K) in ((2808, 2752), (3744, 2752), (6000, 2752), (2808, 1024), (6000, 1024)):
    N = 1024
    A = torch.randn(M, K, device="cuda").to(tdt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tdt)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    big = torch.empty(80 * 1024 * 1024, device="cuda")
    r = {}
    for v in variants:
        ts = []
        try:
            for _ in range(7):
                big.zero_()                              # cold L2 / MALL like inside the block sequence
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.call("toc3d_linear_ex", dt, lib.EPI_RESIDUAL, v, A, K, W, K, b, out, N, res, N, 0, None, None, M, N, K, 0, S_())
                e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        except RuntimeError:
            continue
        r[v] = sorted(ts)[3]
    top = sorted(r.items(), key=lambda kv: kv[1])[:10]
    print(f"M={M} K={K}: " + "  ".join(f"v{v}:{t:.1f}" for v, t in top), flush=True)
