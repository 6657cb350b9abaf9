#!/usr/bin/env python3
"""Round 6: the phased tiles (60-62) and the 128x128 family on the large-M launches of BASELINE.json config 4 / B = 2 -- isolated, warm (back to back) and cold
(behind a 256 MB flush), bf16 and bf16 x 3 on planes, with the frame's real epilogues.  Variants 67-69 / 167 were the PERSISTENT forms of the phased kernel built and
measured in round 6 (one workgroup per CU slot walking its tiles, the next tile's prologue requested in front of the epilogue): +1-2 % on the plain epilogues, -4 ... -12 %
on the statistics epilogues, removed again -- profiles/r06_persistent_phased.txt is this script's output with them in the library (they print n/a now).
    python tools/ubench/persist_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toc3d_amd import lib

DEV = "cuda:0"
S = lib.stream_ptr
flush = torch.empty(64 << 20, dtype=torch.float32, device=DEV)


def planes(t):
    out = torch.empty_like(t)
    lib.call("toc3d_x3_planes", t, t.shape[1], out, out.shape[1], t.shape[0], t.shape[1], S())
    return out


def bench(fn, cold, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def run(prec, M, N, K, epi):
    x3 = prec == "x3"
    tdt = torch.float32 if x3 else torch.bfloat16
    dt = lib.F32X3P if x3 else lib.BF16
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M, K, generator=g)).to(DEV).to(tdt)
    W = (torch.randn((N + 127) // 128 * 128, K, generator=g) * K ** -0.5).to(DEV).to(tdt)
    if x3:
        A, W = planes(A), planes(W)
    bias = torch.randn(N, generator=g).to(DEV)
    c1 = torch.randn(N, generator=g).to(DEV)
    if epi == 7:       # w1|w2: SwiGLU, LayerNorm in (statistics of the projection), statistics out
        Hp = N // 2
        out = torch.empty(M, Hp, dtype=tdt, device=DEV)
        st_in = torch.zeros(4 + M * 16 * 2, device=DEV); st_in[4:] = 1.0
        st_in.view(torch.int32)[0] = 16
        st_out = torch.zeros(4 + M * 44 * 2, device=DEV)
        fn = lambda v: lib.call("toc3d_linear_fused", dt, 7, v, A, K, W, K, bias, out, Hp, None, 0, 0, None, None, M, N, K, Hp - 22, st_out, 44, st_in, 16 | 16 << 32, c1, K, 1e-6, None, 0, None, S())
    elif epi == 5:     # w3: residual, LayerNorm in
        out = torch.zeros(M, N, device=DEV)
        st_in = torch.zeros(4 + M * 44 * 2, device=DEV); st_in[4:] = 1.0
        fn = lambda v: lib.call("toc3d_linear_fused", dt, 5, v, A, K, W, K, bias, out, N, out, N, 0, None, None, M, N, K, 0, None, 0, st_in, 44 | 43 << 32, c1, K - 22, 1e-6, None, 0, None, S())
    elif epi == 6:     # proj: residual + act copy + statistics
        out = torch.zeros(M, N, device=DEV)
        act = torch.empty(M, N, dtype=tdt, device=DEV)
        st_out = torch.zeros(4 + M * 16 * 2, device=DEV)
        fn = lambda v: lib.call("toc3d_linear_fused", dt, 6, v, A, K, W, K, bias, out, N, out, N, 0, None, None, M, N, K, 0, st_out, 16, None, 0, None, 0, 0.0, act, N, None, S())
    else:              # q|k|v: bias
        out = torch.empty(M, N, dtype=tdt, device=DEV)
        fn = lambda v: lib.call("toc3d_linear_ex", dt, 0, v, A, K, W, K, bias, out, N, None, 0, 0, None, None, M, N, K, 0, S())
    fl = 2.0 * M * N * K * (3 if x3 else 1)
    row = []
    for v in (16, 116, 17, 49, 60, 160, 67, 167, 61, 68, 62, 69):
        try:
            w_, c_ = bench(lambda: fn(v), False), bench(lambda: fn(v), True)
            row.append(f"v{v}: {w_:6.1f} / {c_:6.1f} us ({fl / w_ / 1e6:5.0f} TF)")
        except RuntimeError as e:
            row.append(f"v{v}: n/a")
    print(f"{prec} epi{epi} {M}x{N}x{K}  warm / cold (MFMA TF warm{', bf16-equivalent x 3' if x3 else ''}):\n   " + "\n   ".join(row), flush=True)


if __name__ == "__main__":
    for prec in ("bf16", "x3"):
        for M in (24000, 12000, 6000):
            run(prec, M, 5504, 1024, 7)
            run(prec, M, 3072, 1024, 0)
            run(prec, M, 1024, 2752, 5)
            run(prec, M, 1024, 1024, 6)
