#!/usr/bin/env python3
"""GPU development check of the phased big-tile GEMM variants (toc3d_linear_ex variants 60-63):
  1. bit-equality with the 128x128 reference variant over ragged shapes / every K-tile count parity / all epilogues;
  2. race screen: hundreds of launches on fixed inputs with an LDS-heavy kernel co-running on a second stream;
  3. timing on the frame's shapes: warm back-to-back loops and cold single launches (behind a cache-sized memset).
Usage: python tools/gemm_phased_check.py [check] [race] [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toc3d_amd import lib

dev = "cuda:0"
dt, tdt = lib.BF16, torch.bfloat16
S = lib.stream_ptr
PH = [int(v) for v in os.environ.get("PHASED", "60,61,62,63").split(",")]
what = set(sys.argv[1:]) or {"check", "race", "time"}


def mk(M, N, K, epi, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(M, K, device=dev, generator=g).to(tdt)
    Np = (N + 127) // 128 * 128
    W = torch.zeros(Np, K, device=dev, dtype=tdt)
    W[:N] = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(tdt)
    b = torch.randn(N, device=dev, generator=g)
    ldo = N // 2 if epi == lib.EPI_SWIGLU else N
    odt = torch.float32 if epi == lib.EPI_RESIDUAL else tdt
    res = torch.randn(M, N, device=dev, generator=g) if epi == lib.EPI_RESIDUAL else None
    return A, W, b, ldo, odt, res


def run(v, epi, A, W, b, out, ldo, res, M, N, K, rep=None, ridx=None):
    nv = (N // 2 - 22) if epi == lib.EPI_SWIGLU else 0
    lib.call("toc3d_linear_ex", dt, epi, v, A, K, W, K, b, out, ldo, res, N if res is not None else 0, 0, rep, ridx, M, N, K, nv, S())


if "check" in what:
    bad = 0
    for (M, N, K) in [(777, 640, 512), (6000, 1024, 1024), (2178, 3072, 1024), (3744, 5504, 1024), (2808, 1024, 2752), (300, 256, 64), (513, 384, 128),
                      (1000, 1024, 192), (256, 256, 1024), (1, 128, 64), (6000, 256, 2304)]:
        for epi in (lib.EPI_BIAS, lib.EPI_RESIDUAL, lib.EPI_SWIGLU, lib.EPI_GELU):
            if epi == lib.EPI_SWIGLU and N % 64:
                continue
            A, W, b, ldo, odt, res = mk(M, N, K, epi)
            ref = torch.zeros(M, ldo, dtype=odt, device=dev)
            run(16, epi, A, W, b, ref, ldo, res, M, N, K)
            for v in PH + [100 + p for p in PH]:
                out = torch.full((M, ldo), 7.0, dtype=odt, device=dev)
                run(v, epi, A, W, b, out, ldo, res, M, N, K)
                torch.cuda.synchronize()
                if not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
                    d = (out.float() - ref.float()).abs()
                    rows = (d.max(dim=1).values > 0).nonzero().flatten()
                    print(f"MISMATCH v{v} epi{epi} M={M} N={N} K={K}: max abs {d.max().item():.3e}, {rows.numel()} rows differ, first {rows[:8].tolist()}")
                    bad += 1
    # representative-row capture + modular residual rows
    M, N, K = 3744, 1024, 1024
    A, W, b, ldo, odt, res = mk(M, N, K, lib.EPI_RESIDUAL, 3)
    ridx = torch.full((M,), -1, dtype=torch.int32, device=dev)
    ridx[torch.arange(0, M, 77, device=dev)] = torch.arange(len(range(0, M, 77)), dtype=torch.int32, device=dev)
    outs = []
    for v in [16] + PH:
        out = torch.zeros(M, N, device=dev)
        rep = torch.zeros(64, N, device=dev)
        run(v, lib.EPI_RESIDUAL, A, W, b, out, ldo, res, M, N, K, rep, ridx)
        outs.append((out, rep))
    torch.cuda.synchronize()
    for (o, r), v in zip(outs[1:], PH):
        if not (torch.equal(o, outs[0][0]) and torch.equal(r, outs[0][1])):
            print(f"MISMATCH v{v} representative-row capture"); bad += 1
    print("check:", "OK" if bad == 0 else f"{bad} mismatches", flush=True)

if "race" in what:
    M, N, K = 6000, 3072, 1024
    A, W, b, ldo, odt, res = mk(M, N, K, lib.EPI_BIAS, 5)
    side = torch.cuda.Stream()
    # LDS-heavy co-runner: window ranking kernel (uses 1024-thread workgroups and LDS atomics)
    sc = torch.randn(6 * 1000, device=dev)
    order = torch.empty(6, 1000, dtype=torch.int64, device=dev)
    A2, W2, b2, ldo2, odt2, res2 = mk(3000, 1024, 1024, lib.EPI_RESIDUAL, 6)
    o2 = torch.zeros(3000, 1024, device=dev)
    bad = 0
    for v in PH:
        ref = torch.zeros(M, N, dtype=tdt, device=dev)
        run(v, lib.EPI_BIAS, A, W, b, ref, ldo, None, M, N, K)
        torch.cuda.synchronize()
        outs = [torch.zeros(M, N, dtype=tdt, device=dev) for _ in range(8)]
        nbad = 0
        for it in range(40):
            with torch.cuda.stream(side):
                for _ in range(6):
                    lib.call("toc3d_rank_desc", sc, 6, 1000, order, S())
                    run(16, lib.EPI_RESIDUAL, A2, W2, b2, o2, ldo2, res2, 3000, 1024, 1024)
            for o in outs:
                run(v, lib.EPI_BIAS, A, W, b, o, ldo, None, M, N, K)
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o.view(torch.uint8), ref.view(torch.uint8)):
                    nbad += 1
                o.zero_()
        print(f"race screen v{v}: {nbad} differing launches of {40 * 8}", flush=True)
        bad += nbad
    print("race:", "OK" if bad == 0 else "FAILED", flush=True)

if "time" in what:
    C, Hp = 1024, 2752
    shapes = []
    for M in (6000, 3744, 2808, 2178):
        shapes += [("qkv", lib.EPI_BIAS, M, 3072, 1024), ("w12", lib.EPI_SWIGLU, M, 2 * Hp, 1024), ("w3", lib.EPI_RESIDUAL, M, 1024, Hp),
                   ("proj", lib.EPI_RESIDUAL, M, 1024, 1024)]
    variants = [int(v) for v in os.environ.get("VARIANTS", "16,17,45,47,49,114,116,117,126,145").split(",")] + PH + [100 + p for p in PH]
    flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
    for name, epi, M, N, K in shapes:
        A, W, b, ldo, odt, res = mk(M, N, K, epi)
        out = torch.zeros(M, ldo, dtype=odt, device=dev)
        vs = [v for v in variants if not (epi == lib.EPI_SWIGLU and v in (45, 145))]
        warm, cold = {}, {}
        for v in vs:
            run(v, epi, A, W, b, out, ldo, res, M, N, K)
        torch.cuda.synchronize()
        for v in vs:
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(v, epi, A, W, b, out, ldo, res, M, N, K)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            warm[v] = sorted(ts)[2]
            ts = []
            for _ in range(7):
                flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(v, epi, A, W, b, out, ldo, res, M, N, K)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            cold[v] = sorted(ts)[3]
        fl = 2.0 * M * N * K
        best_w, best_c = min(warm, key=warm.get), min(cold, key=cold.get)
        print(f"{name:5s} M={M:5d} N={N:5d} K={K:5d} | warm best v{best_w} {fl / warm[best_w] / 1e9:5.0f}TF | cold best v{best_c} {cold[best_c] * 1e3:5.1f}us {fl / cold[best_c] / 1e9:5.0f}TF || "
              + " ".join(f"v{v}:{fl / warm[v] / 1e9:4.0f}/{fl / cold[v] / 1e9:4.0f}" for v in vs), flush=True)
