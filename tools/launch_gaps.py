#!/usr/bin/env python3
"""Idle time between the kernels of each flow field in a rocprofv3 kernel trace of bench.py.

    python tools/launch_gaps.py gpurun_out/prof/<run>/<pid>_kernel_trace.csv

A field = the kernels between two `convex_upsample_kernel` launches (one per field on the reference's path);
span = end to end, kernel sum = sum of durations, idle = sum of the positive gaps between consecutive kernels."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ups = [i for i, r in enumerate(rows) if "convex_upsample" in r["Kernel_Name"]]
    print("| field | span ms | kernel sum ms | idle ms | launches |\n|---|---|---|---|---|")
    for k, (a, b) in enumerate(zip(ups[:-1], ups[1:])):
        t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
        ks = rows[a + 1:b + 1]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
        idle, prev = 0, t0
        for r in ks:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s > prev:
                idle += s - prev
            prev = max(prev, e)
        print(f"| {k} | {(t1 - t0) / 1e6:.2f} | {busy / 1e6:.2f} | {idle / 1e6:.2f} | {len(ks)} |")


if __name__ == "__main__":
    main()
