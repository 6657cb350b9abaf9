#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/gpu/profile.sh (gpurun_out/{kt,fs,wsz}) into the files committed under profiles/:
  profiles/<tag>_kernel_stats.csv      per-kernel-family totals of the --kernel-trace --stats run
  profiles/<tag>_gemm_hbm_traffic.json memory-side bytes per GEMM launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes
(gfx950 correction from MI355X_MICROARCH.md: read bytes = 2 x FETCH_SIZE[KB]; WRITE_SIZE used as is)."""
import csv, json, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(root, "gpurun_out")     # directory holding kt/ fs/ wsz/
cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 10 --warmup 3 --groups 1 --no-cpu-baseline --no-breakdown --tune-cache tune.json"


def family(name):
    for key in ("gemm_phased_kernel", "gemm_kernel", "attn_rot_x3_kernel", "attn_rot_kernel", "attn_small_kernel", "attn_kernel", "ln_rows_kernel", "ln_act_kernel", "ln_rebase_kernel", "gather_merge_ln_kernel", "scatter_update_kernel",
                "window_topk_kernel", "rank_desc_kernel", "motion_queries_kernel", "collapse_kernel", "score_tokens_kernel", "im2col", "nhwc_to_nchw",
                "abs_pos", "pack_", "window_map_dense", "score_head", "global_mean_half", "copy_segments", "copy_bytes", "prefetch"):
        if key in name:
            return key
    return "other (torch / runtime): " + re.sub(r"\(.*", "", name)[:60]


rows = list(csv.DictReader(open(os.path.join(out, "kt", "kt_kernel_stats.csv"))))
fam = defaultdict(lambda: [0, 0.0])
for r in rows:
    f = fam[family(r["Name"])]
    f[0] += int(r["Calls"])
    f[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in fam.values())
with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
    fo.write("# (every step of the run is in the trace; per-family totals, then the per-instantiation rows of the GEMM)\n")
    fo.write("family,calls,total_us,avg_us,percent\n")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        fo.write(f"\"{k}\",{v[0]},{v[1] / 1e3:.1f},{v[1] / v[0] / 1e3:.2f},{100 * v[1] / tot:.2f}\n")
    fo.write("\nname,calls,total_us,avg_us\n")
    for r in rows:
        if "gemm_kernel" in r["Name"] or "gemm_phased_kernel" in r["Name"] or "gemm_chain_kernel" in r["Name"]:
            fo.write(f"\"{r['Name'][:160]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f}\n")
g = [sum(fam[k][0] for k in ("gemm_kernel", "gemm_phased_kernel", "gemm_chain_kernel")), sum(fam[k][1] for k in ("gemm_kernel", "gemm_phased_kernel", "gemm_chain_kernel"))]
print(f"GEMM: {g[0]} launches, avg {g[1] / g[0] / 1e3:.2f} us, {100 * g[1] / tot:.1f}% of kernel time")


def pmc(dirname, prefix, counter):
    n, s = 0, 0.0
    for r in csv.DictReader(open(os.path.join(out, dirname, f"{prefix}_counter_collection.csv"))):
        if ("gemm_kernel" in r["Kernel_Name"] or "gemm_phased_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            n += 1
            s += float(r["Counter_Value"])
    return n, s / max(n, 1)


def pmc_per_shape(dirname, prefix, counter):
    """(kernel instantiation, grid size) -> [launches, sum]: one group per GEMM shape of the frame (a tile variant serves a few shapes, told apart by the grid)."""
    from tools.summarize_pmc import short
    per = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(out, dirname, f"{prefix}_counter_collection.csv"))):
        if ("gemm_kernel" in r["Kernel_Name"] or "gemm_phased_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            k = per[(short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))]
            k[0] += 1
            k[1] += float(r["Counter_Value"])
    return per


if not os.path.exists(os.path.join(out, "fs", "fs_counter_collection.csv")):
    sys.exit(0)                                           # kernel trace only: no PMC passes in this directory
nf, fetch = pmc("fs", "fs", "FETCH_SIZE")
nw, write = pmc("wsz", "wsz", "WRITE_SIZE")
fs_shape, ws_shape = pmc_per_shape("fs", "fs", "FETCH_SIZE"), pmc_per_shape("wsz", "wsz", "WRITE_SIZE")
js = {"kernel": "gemm_kernel<*> (all toc3d_linear launches of the step)", "launches_in_trace": nf,
      "FETCH_SIZE_KB_avg_per_launch": fetch, "WRITE_SIZE_KB_avg_per_launch": write,
      "correction": "gfx950 rocprofv3: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> read bytes = 2 x FETCH_SIZE "
                    "(MI355X_MICROARCH.md, HBM); WRITE_SIZE uncalibrated, used as is",
      "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
      # per GEMM shape: which launches move how much -- epi5 / epi6 with N = 1024 outputs are attn.proj and mlp.w3 (f32 residual read-modify-write
      # + the bf16 copy of the rows in their epilogues); grid = workgroups of 256 / 512 threads x tiles
      "per_shape": [{"kernel": k[0], "grid": k[1], "launches": v[0], "read_MB": 2 * v[1] / v[0] / 1024, "write_MB": (ws_shape[k][1] / ws_shape[k][0] / 1024) if k in ws_shape else None}
                    for k, v in sorted(fs_shape.items(), key=lambda kv: -kv[1][1])],
      "schedule": __import__("toc3d_amd.backbone", fromlist=["schedule_defaults"]).schedule_defaults("bf16"),   # the launch schedule the passes ran (bench.py quotes the file only while it matches)
      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (then WRITE_SIZE in a separate pass) -- " + cmd}
json.dump(js, open(os.path.join(root, "profiles", f"{tag}_gemm_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(js)[:400])
