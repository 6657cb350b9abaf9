"""Development tool: per-tile timeline of one toc3d_linear_chain launch (toc3d_linear_chain_trace) -- where a tile's time goes (dequeue,
dependency wait, K loop + epilogue, publish), how the XCDs shared the bands, how busy the workgroup slots were.
    python tools/ubench/chain_timeline.py [M] [config] [n_bands] [grid] [flags] [lag]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from toc3d_amd import lib  # noqa: E402
from test_gpu_chain import Block, new_state  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2898
config = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_bands = int(sys.argv[3]) if len(sys.argv) > 3 else 8
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 768
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lag = int(sys.argv[6]) if len(sys.argv) > 6 else 1

b = Block(M, 1024, 2730)
state = new_state()
launch = b.prepare_chain(config, state, n_bands=n_bands, grid=grid, flags=flags, lag=lag)
for _ in range(3):
    launch()
torch.cuda.synchronize()
cap = 8192
buf = torch.zeros(1 + cap * 8, dtype=torch.int64, device="cuda:0")
lib.call("toc3d_linear_chain_trace", buf, cap)
flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda:0")
flush.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
lib.call("toc3d_linear_chain_trace", None, 0)
t = buf.cpu().numpy().astype(np.uint64)
n = int(t[0])
rec = t[1:1 + n * 8].reshape(n, 8)
hdr = rec[:, 0]
xcc = (hdr >> np.uint64(56)).astype(int) - 1
wg = ((hdr >> np.uint64(32)) & np.uint64(0xffffff)).astype(int)
entry = (hdr & np.uint64(0xffffffff)).astype(np.int64)
op = (entry >> 28) & 7
ts = rec[:, 1:6].astype(np.float64) / 100.0            # us
t0 = ts[:, 0].min()
ts -= t0
print(f"M={M} config={config} bands={n_bands} grid={grid} flags={flags} lag={lag}: {n} tiles, launch {e0.elapsed_time(e1) * 1e3:.1f} us (events), "
      f"first dequeue -> last publish {ts[:, 4].max():.1f} us")
names = ["dequeue", "dep wait", "tile", "publish"]
for o in sorted(set(op.tolist())):
    m = op == o
    d = np.diff(ts[m], axis=1)
    print(f"  op {o}: {m.sum():4d} tiles | " + " | ".join(f"{nm} avg {d[:, i].mean():6.2f} max {d[:, i].max():6.2f}" for i, nm in enumerate(names)) +
          f" | starts {ts[m, 2].min():6.1f}..{ts[m, 2].max():6.1f} ends {ts[m, 3].min():6.1f}..{ts[m, 3].max():6.1f}")
tot = np.diff(ts, axis=1).sum(axis=0)
print("  slot time by phase: " + ", ".join(f"{nm} {100 * v / tot.sum():.1f} %" for nm, v in zip(names, tot)))
wgs = len(set(wg.tolist()))
span = ts[:, 4].max()
print(f"  {wgs} workgroups took tiles; busy fraction of their slots over the span: {100 * tot.sum() / (wgs * span):.1f} % (tile phase only: {100 * tot[2] / (wgs * span):.1f} %)")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: {m.sum():4d} tiles, {len(set(wg[m].tolist())):3d} workgroups, M-tiles {sorted(set(((entry[m] >> 16) & 0xfff).tolist()))[:3]}.., last publish {ts[m, 4].max():6.1f} us")
# utilisation over time: tiles in their K loop per 10 us bucket
edges = np.arange(0, span + 10, 10)
act = [(np.minimum(ts[:, 3], hi) - np.maximum(ts[:, 2], lo)).clip(min=0).sum() / 10 for lo, hi in zip(edges[:-1], edges[1:])]
print("  tiles in flight per 10 us: " + " ".join(f"{a:.0f}" for a in act))
