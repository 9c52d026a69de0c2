"""Print the per-layer-shape table of a bench.py JSON line (roofline.all_variants[*].shapes)."""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"{d['value']:.2f} fields/s, {d['ms_per_step']:.2f} ms/field")
r = d.get("roofline")
if r:
    print(r["kernel"], f"frac {r['frac']:.3f}", f"{r['avg_launch_us']:.1f} us")
    rows = []
    for k, v in r["all_variants"].items():
        short = k.replace("conv_gemm_", "").replace("_kernel", "")
        for sk, (n, us, tf) in v.get("shapes", {}).items():
            rows.append((n * us, sk, short, n, us, tf))
    for tot, sk, short, n, us, tf in sorted(rows, reverse=True):
        print(f"  {sk:16s} {short:52s} {n:4d} x {us:7.1f} us  {tf:6.1f} TFLOP/s")
    for k, v in r.get("hbm_kernels", {}).items():
        print(f"  {k}: {v['launches']} x {v['avg_launch_us']:.1f} us")
