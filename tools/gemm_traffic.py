#!/usr/bin/env python3
"""Where do the memory-side bytes of the wide GEMMs go?  Two modes:

    python tools/gemm_traffic.py run                      (under `rocprofv3 --kernel-trace --pmc <counters> -d DIR -o NAME --output-format csv --`)
        launches, REPS times over, for every shape of SHAPES and every XCD order of its shipped tile (v, v + 100, v + 200, v + 300; v % 100 = the table's tile):
        one toc3d_copy_bytes that rewrites the A operand (as the producing kernel of the frame leaves it: freshly written, in some XCD's L2 / the Infinity Cache)
        and then the GEMM.  The launch ORDER is fixed, so the summariser can tell the orders of one kernel instantiation apart.
    python tools/gemm_traffic.py summarize OUT.txt DIR1 [DIR2 ...]
        per (shape, order): memory-side read / write bytes from the request-size counters (32 / 64 / 128 B), L2 hit rate, average fabric read latency
        (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, in L2 clocks), against the algorithmic bytes (A + W read once, output written once).
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (epilogue, M, N, K, tile variant of the shipped 800x320 table): w1|w2 and q|k|v at the dense and a mid accelerated M, w3, proj
SHAPES = [(2, 6000, 5504, 1024, 16), (0, 6000, 3072, 1024, 45), (1, 6000, 1024, 2752, 17), (1, 6000, 1024, 1024, 17), (2, 3744, 5504, 1024, 16), (1, 3744, 1024, 2752, 29)]
ORDERS = (0, 100, 200, 300)
REPS = 6


def run():
    import torch
    from toc3d_amd import lib
    dev = "cuda:0"
    bufs = []
    for epi, M, N, K, v in SHAPES:
        torch.manual_seed(M + N + K)
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        A2 = A.clone()
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        if epi == lib.EPI_RESIDUAL:
            out, ldo, res = torch.zeros(M, N, device=dev), N, torch.randn(M, N, device=dev)
        elif epi == lib.EPI_SWIGLU:
            out, ldo, res = torch.zeros(M, N // 2, dtype=torch.bfloat16, device=dev), N // 2, None
        else:
            out, ldo, res = torch.zeros(M, N, dtype=torch.bfloat16, device=dev), N, None
        bufs.append((A, A2, W, b, out, ldo, res))
    s = lib.stream_ptr()
    for _ in range(REPS):
        for (epi, M, N, K, v), (A, A2, W, b, out, ldo, res) in zip(SHAPES, bufs):
            for o in ORDERS:
                lib.call("toc3d_copy_bytes", A, A2, A.numel() * 2, s)
                lib.call("toc3d_linear_ex", lib.BF16, epi, v + o, A, K, W, K, b, out, ldo, res, N if res is not None else 0, 0, None, None, M, N, K,
                         (N // 2 - 22) if epi == lib.EPI_SWIGLU else 0, s)
    torch.cuda.synchronize()
    print("launched", REPS, "x", len(SHAPES), "shapes x", len(ORDERS), "orders")


def summarize(out_path, dirs):
    seq = [(sh, o) for _ in range(REPS) for sh in SHAPES for o in ORDERS]
    acc = {}                                              # (shape, order) -> counter -> [n, sum]
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if "gemm_kernel" in r["Kernel_Name"] or "gemm_phased_kernel" in r["Kernel_Name"]]
        by_disp = {}
        for r in rows:
            by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp = sorted(by_disp)
        assert len(disp) == len(seq), (d, len(disp), len(seq))
        for did, key in zip(disp, seq):
            for c, val in by_disp[did].items():
                e = acc.setdefault(key, {}).setdefault(c, [0, 0.0])
                e[0] += 1
                e[1] += val
    lines = ["# python tools/gemm_traffic.py (rocprofv3 --pmc passes; per launch averages; A rewritten by a copy kernel in front of every launch, W re-read from wherever the",
             "# previous launch of the same shape left it: L2 of the XCD / Infinity Cache -- the state the frame's launches find their operands in).",
             "# order 0 = XCD chunks of row-major tiles, 100 = 8 row bands (an XCD keeps its A band, streams ALL of W), 200 = 4 row bands x 2 column halves, 300 = 2 x 4.",
             "# read MB = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B (memory-side requests of the eight L2s: Infinity Cache AND HBM -- the L2 cannot tell them apart);",
             "# latency = TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ in L2 clocks (an HBM miss costs ~900 shader clocks, MI355X_MICROARCH.md: a low average = served by the Infinity Cache)."]
    for (epi, M, N, K, v) in SHAPES:
        alg_r = (M * K + N * K) * 2 + (M * N * 4 if epi == 1 else 0)
        alg_w = M * N * 4 if epi == 1 else (M * (N // 2) * 2 if epi == 2 else M * N * 2)
        lines.append(f"epi{epi} M={M} N={N} K={K} tile v{v}: algorithmic read {alg_r / 1e6:.1f} MB, write {alg_w / 1e6:.1f} MB")
        for o in ORDERS:
            c = {k: s_ / n for k, (n, s_) in acc.get(((epi, M, N, K, v), o), {}).items()}
            g = lambda k: c.get(k, float("nan"))
            rd = (32 * g("TCC_EA0_RDREQ_32B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum") + 64 * (g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum") - g("TCC_EA0_RDREQ_128B_sum"))) / 1e6
            wr = (64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))) / 1e6
            hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if c.get("TCC_HIT_sum") is not None else float("nan")
            lat = g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum") if c.get("TCC_EA0_RDREQ_sum") else float("nan")
            lines.append(f"    order {o:3d}: read {rd:7.1f} MB ({rd / (alg_r / 1e6):4.2f}x)  write {wr:6.1f} MB  L2 hit {100 * hit:5.1f} %  L2 requests {g('TCC_REQ_sum') / 1e6:6.2f} M  "
                         f"fabric read latency {lat:6.0f} clk  DRAM-destined reads {g('TCC_EA0_RDREQ_DRAM_sum') / max(g('TCC_EA0_RDREQ_sum'), 1):4.2f} of all")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3:])
    else:
        run()
