#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of a HIP source file, as hipcc reports them
(-Rpass-analysis=kernel-resource-usage).  A non-zero scratch size means the accumulators spill.

    python tools/kernel_resources.py video-flow-ml_amd/vfml/csrc/conv_gemm_split.hip [filter]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
                        src, "-o", "/dev/null"] + sys.argv[3:], capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    if not blocks:
        sys.stderr.write(r.stderr[-2000:])
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.split("\n")
    for b, d in zip(blocks, dem):
        d = d.replace("(anonymous namespace)::", "").replace("(SplitArgs)", "")
        if flt not in d:
            continue

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        print("%-78s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
            d[:78], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
            g("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
