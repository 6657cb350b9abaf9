#!/usr/bin/env python3
"""Run the window-attention kernels on the shapes of the ToC3D_faster frame (for timing / rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toc3d_amd import lib, synth
dev = "cuda:0"
C, heads = 1024, 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def run(nwin, n, L, dense_pads=0, label=""):
    M = nwin * n
    qkv = (torch.randn(M, 3 * C, device=dev)).to(torch.bfloat16)
    out = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    rows = torch.arange(M, dtype=torch.int32, device=dev).reshape(nwin, n).contiguous()
    slots = torch.stack([torch.randperm(L * L, device=dev)[:n] for _ in range(nwin)]).int().contiguous()
    cnt = torch.full((nwin,), n, dtype=torch.int32, device=dev)
    npad = torch.full((nwin,), dense_pads, dtype=torch.int32, device=dev) if dense_pads else None
    cos, sin = synth.rope_tables(L)
    cos, sin = cos.to(dev), sin.to(dev)
    vb = torch.randn(C, device=dev)
    def call():
        lib.call("toc3d_window_attention", lib.BF16, qkv, 3 * C, out, C, rows, slots, cnt, None, npad, None, n, nwin, n, heads, cos, sin, L,
                 vb if dense_pads else None, 0.125, lib.stream_ptr())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:28s} nwin={nwin:3d} n={n:3d}: {1e3 * e0.elapsed_time(e1) / reps:7.1f} us", flush=True)
run(48, 129, 16, label="accel win stage1 (small)")
run(48, 77, 16, label="accel win stage3 (small)")
run(18, 201, 20, label="accel glb stage1 (small)")
run(48, 256, 16, dense_pads=5, label="dense win (flash)")
run(18, 400, 20, dense_pads=5, label="dense glb (flash)")
