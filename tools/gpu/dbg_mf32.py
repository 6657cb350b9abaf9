import sys, os
sys.path.insert(0, os.getcwd())
import torch
from toc3d_amd import lib
dev = "cuda:0"
M, N, K = 128, 128, 64
v = int(sys.argv[1]) if len(sys.argv) > 1 else 72
A = torch.zeros(M, K); A[:, 0] = torch.arange(1, M + 1).float()
W = torch.zeros(N, K); W[:, 0] = torch.arange(1, N + 1).float() / 128.0
a = A.to(dev).bfloat16(); w = W.to(dev).bfloat16()
out = torch.zeros(M, N, device=dev)
lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_RESIDUAL, v, a, K, w, K, None, out, N, None, 0, 0, None, None, M, N, K, 0, lib.stream_ptr())
torch.cuda.synchronize()
o = out.cpu()
ref = A[:, :1] @ W[:, :1].T
print("max err", (o - ref).abs().max().item())
# decode: value = m' * n' / 128 ; find m', n' for a few positions
bad = (o - ref).abs() > 1e-3
print("bad count", int(bad.sum()), "of", M * N)
for (m, n) in [(0, 0), (0, 1), (0, 4), (0, 8), (0, 16), (1, 0), (16, 0), (32, 0), (5, 37), (70, 90)]:
    val = o[m, n].item() * 128
    # find factorization candidates with m',n' in 1..128
    c = [(mm, int(round(val / mm))) for mm in range(1, 129) if abs(val / mm - round(val / mm)) < 1e-6 and 1 <= round(val / mm) <= 128]
    print((m, n), "got", val, "want", (m + 1) * (n + 1), "cands", c[:6])
# test k mapping: A row 0 has value at k = kk, W row 0 at k = kk -> out[0,0] = 1 for every kk
for kk in (1, 7, 8, 15, 16, 31, 32, 63):
    A2 = torch.zeros(M, K); A2[:, kk] = 1; W2 = torch.zeros(N, K); W2[:, kk] = 1
    out.zero_()
    lib.call("toc3d_linear_ex", lib.BF16, lib.EPI_RESIDUAL, v, A2.to(dev).bfloat16(), K, W2.to(dev).bfloat16(), K, None, out, N, None, 0, 0, None, None, M, N, K, 0, lib.stream_ptr())
    torch.cuda.synchronize()
    print("k", kk, "all ones:", bool((out == 1).all()), "sum", out.sum().item())
