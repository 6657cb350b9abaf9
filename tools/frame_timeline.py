"""Development tool: cut replayed frames out of a rocprofv3 --kernel-trace CSV of `python bench.py ...` and say where a frame's time goes --
launches per kernel family, first start -> last end, sum of kernel durations, idle gaps on the union of all lanes, kernels that are not ours.
A frame = the launches between two toc3d copy_segments launches (the staging of the per-frame inputs is the first launch of every frame).
    python tools/frame_timeline.py gpurun_out/kt/kt_kernel_trace.csv [frame index from the end, default -3 -5]
profiles/r03_where_time_goes*.txt are this tool's output."""
import csv
import re
import sys
from collections import defaultdict

def short(name, side=False):
    m = re.search(r"[a-z][a-z0-9_]*?_kernel", name)                       # plain or mangled (_ZN12_GLOBAL__N_111gemm_kernelI...)
    base = m.group(0) if m else name.split("(")[0].split("<")[0].strip()
    if base.startswith("gemm_"):
        return "gemm_kernel (all toc3d_linear* launches)"
    return base + (" (side lane)" if side else "")


def ours(name):
    return "toc3d" in name or "anonymous namespace" in name or "_kernel" in name and "rocclr" not in name and "at::" not in name


def cut_frames(rows):
    """rows: (start ns, end ns, kernel name[, stream id]).  Frames = the launches from one copy_segments launch up to the next one (the last, open one is dropped)."""
    rows = sorted(rows)
    cuts = [i for i, r in enumerate(rows) if "copy_segments" in r[2]]
    return [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]


def summarize(fr):
    """One frame -> launches, span, idle time on the union of all lanes (ns), per-family [launches, ns], kernels that are not ours.
    A launch is on a side lane when its stream is not the stream of the frame's first launch (rows without a stream id: all on one lane)."""
    t0, t1 = fr[0][0], max(r[1] for r in fr)
    main = fr[0][3] if len(fr[0]) > 3 else None
    busy_end, idle = fr[0][1], 0
    for s, e, *_ in fr[1:]:
        if s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end, e)
    fam = defaultdict(lambda: [0, 0])
    for s, e, n, *st in fr:
        k = short(n, bool(st) and st[0] != main)
        fam[k][0] += 1
        fam[k][1] += e - s
    return dict(launches=len(fr), span=t1 - t0, idle=idle, families=dict(fam), foreign=sum(1 for r in fr if not ours(r[2])))


def main():
    path = sys.argv[1]
    which = [int(a) for a in sys.argv[2:]] or [-3, -5]
    with open(path) as f:
        rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0")) for r in csv.DictReader(f)]
    frames = cut_frames(rows)
    print(f"# {len(rows)} dispatches, {len(frames)} frames between copy_segments launches; frames taken from the end of the run (the timed region)")
    for w in which:
        r = summarize(frames[w])
        tot = sum(v[1] for v in r["families"].values())
        print(f"\nframe {w}: {r['launches']} launches, first start -> last end {r['span'] / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms "
              f"(side lanes overlap the block chain), idle gaps on the union of all lanes {r['idle'] / 1e3:.1f} us; kernels that are not toc3d kernels: {r['foreign']}")
        side = [v for k, v in r["families"].items() if k.endswith("(side lane)")]
        print(f"on the frame's own lane: {r['launches'] - sum(v[0] for v in side)} launches, {(tot - sum(v[1] for v in side)) / 1e6:.3f} ms of kernel time; "
              f"beside it: {sum(v[0] for v in side)} launches, {sum(v[1] for v in side) / 1e6:.3f} ms")
        print(f"{'family':52s} {'launches':>8s} {'total us':>10s} {'avg us':>8s}")
        for k, (c, t) in sorted(r["families"].items(), key=lambda kv: -kv[1][1]):
            print(f"{k:52s} {c:8d} {t / 1e3:10.1f} {t / 1e3 / c:8.2f}")


if __name__ == "__main__":
    main()
