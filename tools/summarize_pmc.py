#!/usr/bin/env python3
"""Average the counters of rocprofv3 --pmc passes per kernel instantiation.

    python tools/summarize_pmc.py <out.txt> <header comment> <dir1> [<dir2> ...]

Every <dir> is the -d directory of one `rocprofv3 --kernel-trace --pmc ... --output-format csv` pass (separate passes per
counter group, never combined with other tracing: gpurun refuses that).  Launches are grouped by (kernel family, template
arguments that identify the tile, grid size, VGPRs); one line per group and pass with the per-launch averages.  Derived on the
SQ pass that has them: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / n_xcd_se ... ) is NOT derived here -- the raw
counters are kept and the formula is quoted in the header by the caller, so nothing depends on gfx94x fallback formulas
(MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    # rocprofv3 leaves the anonymous-namespace templates mangled: _ZN12_GLOBAL__N_111gemm_kernelIDF16bLi4ELi128E...
    m = re.search(r"gemm_kernelI(DF16b|f)((?:Li\d+E)+)", name)
    if m:
        v = re.findall(r"Li(\d+)E", m[2])
        return f"gemm<{'bf16' if m[1] == 'DF16b' else 'f32'},epi{v[0]},{v[1]}x{v[2]},stages{v[3]},rb{v[4]},waves{v[5]}x{v[6]},occ{v[7]}>"
    m = re.search(r"gemm_phased_kernelI((?:Li\d+E)+)", name)
    if m:
        v = re.findall(r"Li(\d+)E", m[1])
        return f"gemm_phased<epi{v[0]},{v[1]}x{v[2]},waves{v[3]}x{v[4]}>"
    m = re.search(r"attn_small_kernelI(DF16b|f)Li(\d+)E", name)
    if m:
        return f"attn_small<{'bf16' if m[1] == 'DF16b' else 'f32'},maxsub{m[2]}>"
    m = re.search(r"attn_kernelI(DF16b|f)", name)
    if m:
        return f"attn_flash<{'bf16' if m[1] == 'DF16b' else 'f32'}>"
    m = re.search(r"_GLOBAL__N_1\d+([a-z_0-9]+_kernel)", name)
    if m:
        return m[1]
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm_kernel<([^,]+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", n)
    if m:
        return f"gemm<{m[1].replace('__hip_bfloat16', 'bf16')},epi{m[2]},{m[3]}x{m[4]},stages{m[5]},rb{m[6]},waves{m[7]}x{m[8]}>"
    m = re.match(r"gemm_phased_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", n)
    if m:
        return f"gemm_phased<epi{m[1]},{m[2]}x{m[3]},waves{m[4]}x{m[5]}>"
    return re.sub(r"\(.*", "", n)[:60]


def main():
    out, header, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = [f"# {l}" for l in header.split("\\n")]
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            lines.append(f"# {d}: no counter_collection.csv")
            continue
        per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        dur = defaultdict(lambda: [0, 0.0])
        seen = set()
        for r in csv.DictReader(open(files[0])):
            name = short(r["Kernel_Name"])
            if not any(k in name for k in ("gemm", "attn", "ln_", "gather", "scatter")):
                continue
            key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
            c = per[key][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
            did = r.get("Dispatch_Id")
            if did not in seen and "Start_Timestamp" in r:
                seen.add(did)
                dur[key][0] += 1
                dur[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        lines.append(f"# pass {os.path.basename(d.rstrip('/'))}")
        for key, cs in sorted(per.items()):
            n = max(v[0] for v in cs.values())
            parts = [f"{key[0]} grid={key[1]} vgpr={key[2]} lds={key[3]} launches={n}"]
            if dur[key][0]:
                parts.append(f"duration_us={dur[key][1] / dur[key][0]:.2f}")
            parts += [f"{k}={v[1] / v[0]:.4g}" for k, v in sorted(cs.items())]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" not in cs:
                pass
            lines.append(",".join(parts))
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
