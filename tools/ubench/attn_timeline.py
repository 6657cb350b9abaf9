"""Development tool: per-workgroup phase timeline of toc3d_window_attention_rot on MI355X.

Builds a private copy of csrc/attention_rot.hip with -DTOC3D_ATTN_TRACE (libtoc3d_attn_trace.so; the shipped library carries no stamps) and runs
the window shapes of the ToC3D_faster frame on a ViT-L sized q|k|v buffer: dense 16x16 / 20x20 windows and the accelerated blocks' k + 1 keys.
Per launch: the span of the kernel and the median time a workgroup spends in each phase
  entry -> indices back -> operands landed (DMA + barrier) -> compute done -> stores acknowledged,
with and without the weight-prefetch rows (25 MB) riding on the launch.     python tools/ubench/attn_timeline.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "toc3d_amd", "csrc")
SO = os.path.join(ROOT, "tools", "ubench", "bin", "libtoc3d_attn_trace.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(CS, f) for f in ("attention_rot.hip", "capi.cpp", "plan.cpp")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) > os.path.getmtime(s) for s in srcs):
        return
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math", "-DTOC3D_ATTN_TRACE",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-x", "hip"] + srcs + ["-o", SO]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return build()
    build()
    L = ctypes.CDLL(SO)
    dev = "cuda:0"
    C, heads, V, h, w = 1024, 16, 6, 20, 50
    M = V * h * w
    from toc3d_amd import lib
    qkv = torch.randn(M, 3 * C, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, C, dtype=torch.bfloat16, device=dev)
    vb = torch.randn(C, device=dev)
    weights = torch.randn(25 * 1024 * 1024 // 4, device=dev)
    trace = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
    P, I64 = ctypes.c_void_p, ctypes.c_int64
    fn = L.toc3d_window_attention_rot
    fn.argtypes = [ctypes.c_int, P, I64, P, I64, P, P, P, P, P, P, I64, I64, I64, I64, P, I64, P, P, I64, P]
    s = torch.cuda.current_stream().cuda_stream

    def run(name, rows, count, npad, stride, nwin, maxq, pf):
        ptrs = (ctypes.c_void_p * 1)(weights.data_ptr())
        nb = (ctypes.c_int64 * 1)(weights.numel() * 4)
        args = (1, qkv.data_ptr(), 3 * C, out.data_ptr(), C, rows.data_ptr(), None, count.data_ptr(), None, npad.data_ptr() if npad is not None else None, None,
                stride, nwin, maxq, heads, vb.data_ptr() if npad is not None else None, 1 if pf else 0, ptrs, nb, 192, s)
        for _ in range(3):
            assert fn(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); fn(*args); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        trace.zero_()
        assert L.toc3d_attn_trace_set(ctypes.c_void_p(trace.data_ptr())) == 0
        fn(*args)
        torch.cuda.synchronize()
        L.toc3d_attn_trace_set(None)
        t = trace.cpu().numpy()
        live = t[:, 0] > 0
        t = t[live]
        att = t[t[:, 2] > 0]                     # attention workgroups reached the compute phase
        pfw = t[(t[:, 2] == 0) & (t[:, 4] > 0)]  # prefetch workgroups
        t0 = t[:, 0].min()
        end = max(att[:, 4].max(), pfw[:, 4].max() if len(pfw) else 0)
        med = lambda a: float(np.median(a)) * 0.01
        line = (f"{name:28s} pf={int(pf)} | event {min(ts):6.1f} us | span {(end - t0) * 0.01:6.1f} | attention WGs {len(att):4d}: start p50 {med(att[:, 0] - t0):5.1f} max {(att[:, 0].max() - t0) * 0.01:5.1f}"
                f" | idx {med(att[:, 1] - att[:, 0]):5.1f} | operands {med(att[:, 2] - att[:, 1]):5.1f} | compute {med(att[:, 3] - att[:, 2]):5.1f} | store ack {med(att[:, 4] - att[:, 3]):5.1f}"
                f" | WG total p10 {np.percentile(att[:, 4] - att[:, 0], 10) * 0.01:5.1f} med {med(att[:, 4] - att[:, 0]):5.1f} p90 {np.percentile(att[:, 4] - att[:, 0], 90) * 0.01:5.1f} max {(att[:, 4] - att[:, 0]).max() * 0.01:5.1f}"
                f" | compute p90 {np.percentile(att[:, 3] - att[:, 2], 90) * 0.01:5.1f} max {(att[:, 3] - att[:, 2]).max() * 0.01:5.1f} | last attention WG ends {(att[:, 4].max() - t0) * 0.01:6.1f}")
        if len(pfw):
            line += f" | {len(pfw)} prefetch WGs: med {med(pfw[:, 4] - pfw[:, 0]):5.1f}, last ends {(pfw[:, 4].max() - t0) * 0.01:6.1f}"
        print(line, flush=True)

    i32 = dict(dtype=torch.int32, device=dev)
    for Lw in (16, 20):
        nW, N = V * (-(-h // Lw)) * (-(-w // Lw)), Lw * Lw
        rows, slots = torch.empty(nW, N, **i32), torch.empty(nW, N, **i32)
        count, npad = torch.empty(nW, **i32), torch.empty(nW, **i32)
        lib.call("toc3d_window_map_dense", V, h, w, Lw, rows, slots, count, npad, s)
        for pf in (False, True):
            run(f"dense {Lw}x{Lw} ({N} keys)", rows, count, npad, N, nW, int(count.max()), pf)
    g = torch.Generator().manual_seed(0)
    for nwin, n in ((48, 129), (48, 103), (48, 77), (18, 201), (18, 161), (18, 121)):
        # compact rows of an accelerated block: window i owns n consecutive rows
        rows = (torch.arange(nwin * n, dtype=torch.int32) % M).reshape(nwin, n).to(dev)
        count = torch.full((nwin,), n, **i32)
        for pf in (False, True):
            run(f"accel {nwin} windows x {n} keys", rows, count, None, n, nwin, n, pf)


if __name__ == "__main__":
    main()
