"""Development estimate: what would split-K buy on the N=1024 residual GEMMs?  A split-S GEMM does the work of a GEMM with
S x the rows and K / S (same workgroup count and per-workgroup work, minus the fix-up)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
S_ = lib.stream_ptr
dt, tdt = lib.BF16, torch.bfloat16
def best(M, N, K, variants=(8, 9, 10, 13, 14, 16, 17, 24, 26, 28, 29, 1)):
    A = torch.randn(M, K, device="cuda").to(tdt); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tdt)
    b = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    r = {}
    for v in variants:
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.call("toc3d_linear_ex", dt, lib.EPI_RESIDUAL, v, A, K, W, K, b, out, N, res, N, 0, None, None, M, N, K, 0, S_())
            e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        r[v] = sorted(ts)[3]
    v = min(r, key=r.get)
    return v, r[v]
for (M, K) in ((2808, 2752), (3744, 2752), (6000, 2752), (2808, 1024), (3744, 1024), (6000, 1024)):
    line = f"M={M} N=1024 K={K}:"
    for S in (1, 2, 4):
        Ks = (K // S + 63) // 64 * 64
        v, t = best(M * S, 1024, Ks)
        line += f"  S={S}: {t:6.1f} us (v{v})"
    print(line, flush=True)
