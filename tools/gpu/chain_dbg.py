import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from test_gpu_chain import Block, new_state
from toc3d_amd import lib
torch.set_printoptions(linewidth=250)
M, C, Hd = 777, 384, 300
config = 10
b = Block(M, C, Hd)
b.run_separate(1)
ref = b.snapshot()
state = new_state()
for kw in (dict(n_bands=8, grid=768, flags=0), dict(n_bands=1, grid=16, flags=0), dict(n_bands=1, grid=1, flags=0), dict(n_bands=8, grid=768, flags=1)):
    b.fresh()
    b.run_chain(config, state, **kw)
    torch.cuda.synchronize()
    got = b.snapshot()
    d = (got[0] - ref[0])
    nz = d.nonzero()
    print(kw, "x differs at", nz.shape[0], "hid equal", torch.equal(got[2], ref[2]), "stats equal", torch.equal(got[4], ref[4]))
    if nz.shape[0] == 0:
        continue
    rows = torch.bincount(nz[:, 0], minlength=M)
    cols = torch.bincount(nz[:, 1], minlength=C)
    print(" rows with differences per 128-row tile:", [(int((rows[t*128:(t+1)*128] > 0).sum())) for t in range((M + 127) // 128)])
    print(" per-column counts (nonzero):", {int(c): int(cols[c]) for c in cols.nonzero().flatten()[:60]})
    r0 = int(nz[0, 0]); 
    print(" row", r0, "cols 0..7: got", got[0][r0, :8].tolist(), "\n   ref", ref[0][r0, :8].tolist(), "\n   x0 ", b.x0[r0, :8].tolist())
    dl = (got[0] - b.x0)[r0, :8]; dr = (ref[0] - b.x0)[r0, :8]
    print("   delta got", dl.tolist(), "\n   delta ref", dr.tolist(), "\n   c1", b.c1[:8].tolist(), "\n   c2", b.c2[:8].tolist())
    rr = rows.nonzero().flatten()
    print(" rows (first 40):", rr[:40].tolist(), "... last:", rr[-10:].tolist())
