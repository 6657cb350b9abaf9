import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib
S = lib.stream_ptr
M, N, K = 6016, 3072, 1024
def t(A, W, b, v, epi=lib.EPI_BIAS, R=20):
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3): lib.call("toc3d_linear_ex", lib.BF16, epi, v, A, K, W, K, b, out, N, None, 0, 0, None, None, M, N, K, 0, S())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R): lib.call("toc3d_linear_ex", lib.BF16, epi, v, A, K, W, K, b, out, N, None, 0, 0, None, None, M, N, K, 0, S())
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    return us, 2.0 * M * N * K / us / 1e6
b = torch.zeros(N, device="cuda")
for name, mk in (("zeros", lambda *s: torch.zeros(*s, device="cuda")), ("randn", lambda *s: torch.randn(*s, device="cuda")), ("randn*0.01", lambda *s: torch.randn(*s, device="cuda") * 0.01)):
    A = mk(M, K).to(torch.bfloat16); W = mk(N, K).to(torch.bfloat16)
    for v in (16, 17, 8, 1, 116):
        us, tf = t(A, W, b, v)
        print(f"{name:10s} variant {v:3d}: {us:7.1f} us  {tf:6.0f} TF", flush=True)
