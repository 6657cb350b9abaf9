"""Development sweep (round 5): deterministic split-K variants (toc3d_linear_fused_ws) against the shipped table's pick on the frame's N = 1024 residual
GEMMs -- attn.proj (EPI_RESIDUAL_STATS, K = 1024) and mlp.w3 (EPI_RESIDUAL_LN, K = 2752) at every M of ToC3D_faster 6x800x320.  Single cold launches
(a cache-sized memset in front of each, like the autotuner), lower quartile of 9, event timed."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from toc3d_amd import lib

S_ = lib.stream_ptr
dt, tdt = lib.BF16, torch.bfloat16
DEV = "cuda"
C, Hp, Hd = 1024, 2752, 2730
table = {tuple(k): v for k, v in json.load(open(os.path.join(os.path.dirname(lib.LIB_PATH), "tuned", "toc3d_faster_320x800_bf16.json")))["table"]}
SK_TILES = (1, 9, 10, 14, 16, 17, 19, 22, 26, 28, 29, 55, 56)
Ms = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6000, 3744, 3618, 3276, 2898, 2808, 2178]
flush = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=DEV)
ws = torch.zeros(64 * 1024 * 1024, dtype=torch.int32, device=DEV)             # 256 MB: any split of any shape here

print("# cold single launches, us (lower quartile of 9); 'table' = the shipped pick of toc3d_amd/tuned/toc3d_faster_320x800_bf16.json")
for epi, K in ((lib.EPI_RESIDUAL_STATS, C), (lib.EPI_RESIDUAL_LN, Hp)):
    for M in Ms:
        N = C
        A = torch.randn(M, K, device=DEV).to(tdt)
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).to(tdt)
        b = torch.randn(N, device=DEV)
        x = torch.randn(M, N, device=DEV)
        a_raw = torch.empty(M, N, dtype=tdt, device=DEV)
        st_out = torch.zeros(4 + M * 16 * 2, device=DEV)
        slots = (2 * Hp + 127) // 128
        st_in = torch.rand(4 + M * 48 * 2, device=DEV)
        c1 = torch.randn(N, device=DEV)
        if epi == lib.EPI_RESIDUAL_STATS:
            fused = (st_out, 16, None, 0, None, 0, 0.0, a_raw, N, None)
        else:
            fused = (None, 0, st_in, 48 | slots << 32, c1, Hd, 1e-6, None, 0, None)

        def t_of(v):
            args = (dt, epi, v, A, K, W, K, b, x, N, x, N, 0, None, None, M, N, K, 0, *fused)
            ts = []
            try:
                for _ in range(10):
                    flush.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if v >= 1000:
                        lib.call("toc3d_linear_fused_ws", *args, ws, ws.numel() * 4, S_())
                    else:
                        lib.call("toc3d_linear_fused", *args, S_())
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
            except RuntimeError:
                return float("inf")
            return sorted(ts[1:])[2]

        pick = table.get((epi, M, N, K), 16)
        t_pick = t_of(pick)
        flops = 2.0 * M * N * K
        res = sorted((t_of(1000 * s + tv), 1000 * s + tv) for s in (2, 3, 4) for tv in SK_TILES)
        unsplit = sorted((t_of(v), v) for v in (16, 17, 28, 29, 56, 114, 116, 117, 126, 155, 156, 9, 14, 26))
        best = res[0]
        print(f"epi{epi} M={M:5d} K={K:4d} | table v{pick:<4d} {t_pick:6.1f} ({flops / t_pick * 1e-6:4.0f} TF) | best unsplit v{unsplit[0][1]:<4d} {unsplit[0][0]:6.1f} | split-K: "
              + "  ".join(f"v{v}:{t:.1f}" for t, v in res[:6]) + f"   -> {best[0] / min(t_pick, unsplit[0][0]):.2f}x of the best unsplit ({flops / best[0] * 1e-6:4.0f} TF)", flush=True)
