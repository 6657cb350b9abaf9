export PYTHONUNBUFFERED=1
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-batched --no-parity-path --no-other-configs --no-ab --no-calibration"
for L in libtoc3d_prev.so libtoc3d_gfx950.so; do
  TOC3D_LIB=$L $B 2> gpurun_out/r4_sdma_$L.err > /dev/null
done
python - <<'PY'
import re
def load(f):
    d={}
    for ln in open(f):
        m=re.match(r'\s+(toc3d_\S+)\s+([\d.]+) us x\s+(\d+)\s+=\s+([\d.]+) ms',ln)
        if m: d[m.group(1)]=(float(m.group(2)),int(m.group(3)))
    return d
a=load('gpurun_out/r4_sdma_libtoc3d_prev.so.err'); b=load('gpurun_out/r4_sdma_libtoc3d_gfx950.so.err')
tot=0
for k in a:
    if k in b:
        d=(b[k][0]-a[k][0])*a[k][1]
        tot+=d
        if abs(d)>4: print(f"{k:75s} prev {a[k][0]:6.1f} new {b[k][0]:6.1f} x{a[k][1]}  delta {d:+.0f} us/frame")
print('total delta us/frame', round(tot))
for k in b:
    if k not in a: print('only new', k, b[k])
for k in a:
    if k not in b: print('only prev', k, a[k])
PY
