"""Development micro-benchmark: how much slower is a GEMM inside the block sequence than back to back with itself?"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
S = lib.stream_ptr
M, C = 6000, 1024
dt, tdt = lib.BF16, torch.bfloat16
x = torch.randn(M, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
a = torch.empty(M, C, dtype=tdt, device="cuda")
Ws = [(torch.randn(3072, C, device="cuda") * C ** -0.5).to(tdt) for _ in range(8)]   # a different layer's weights every time
bq = torch.randn(3072, device="cuda")
qkv = torch.empty(M, 3072, dtype=tdt, device="cuda")
big = torch.empty(64 * 1024 * 1024, device="cuda")      # 256 MB: flushes L2 + MALL when touched
def ln(): lib.call("toc3d_layernorm_rows", dt, x, C, None, None, g, b, 1e-6, a, C, M, C, S())
def gemm(v, W): lib.call("toc3d_linear_ex", dt, lib.EPI_BIAS, v, a, C, W, C, bq, qkv, 3072, None, 0, 0, None, None, M, 3072, C, 0, S())
def timed(fn_before, v, R=24, same_w=False):
    ts = []
    for i in range(R):
        fn_before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gemm(v, Ws[0] if same_w else Ws[i % 8]); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
ln(); torch.cuda.synchronize()
for v in (16, 17, 117, 116, 8):
    r = [timed(lambda: None, v, same_w=True), timed(lambda: None, v), timed(ln, v), timed(lambda: (ln(), torch.cuda.synchronize()), v),
         timed(lambda: big.zero_(), v), timed(lambda: (big.zero_(), ln()), v)]
    print(f"v{v:3d}: back-to-back same W {r[0]:6.1f} us | rotating W {r[1]:6.1f} | after LN {r[2]:6.1f} | after LN+sync {r[3]:6.1f} | after 256MB memset {r[4]:6.1f} | memset+LN {r[5]:6.1f}", flush=True)
