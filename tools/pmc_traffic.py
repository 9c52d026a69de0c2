#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs - the TCC block has
no room for both) over the same command -> profiles/<tag>_hbm_traffic.json, the file bench.py's `roofline.traffic` quotes.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_hbm_traffic.json "<command>"

hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reads exactly half the bytes of
16-byte streaming loads (LDS-DMA included), WRITE_SIZE is exact for 16-byte stores (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name.strip())
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\s*\[clone[^\]]*\]$", "", name)
    depth, end = 0, len(name)
    for i, ch in enumerate(name):          # cut the argument list: the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    return name[:end].strip()


def collect(root, counter):
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*.csv"), recursive=True):
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            if not rd.fieldnames or "Counter_Name" not in rd.fieldnames or "Kernel_Name" not in rd.fieldnames:
                continue
            per_dispatch = {}
            for row in rd:
                if row["Counter_Name"] != counter:
                    continue
                key = (row.get("Dispatch_Id") or row.get("Correlation_Id"), row["Kernel_Name"])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            for (_, k), v in per_dispatch.items():
                d = acc.setdefault(short(k), [0, 0.0])
                d[0] += 1
                d[1] += v
    return acc


def main():
    froot, wroot, out = sys.argv[1:4]
    how = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = collect(froot, "FETCH_SIZE"), collect(wroot, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw)
        if not n:
            continue
        fk, wk = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
        kernels[k] = {"launches": n, "fetch_size_kb": round(fk, 1), "write_size_kb": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    with open(out, "w") as fo:
        json.dump({"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- "
                           + how + "; per kernel: mean over its dispatches; hbm_bytes_per_launch = (2 * FETCH_SIZE + "
                           "WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes of 16-byte streaming "
                           "loads on gfx950, WRITE_SIZE is exact for 16-byte stores). Kernels with narrower accesses "
                           "(corr_lookup: 4-byte gathers, 8-byte stores) are outside that calibration.",
                   "kernels": kernels}, fo, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:90]:90s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
