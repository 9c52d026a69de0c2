"""Lists the sites of the pattern behind the two-stream lookup finding (profiles/r02_kernel_anatomy.md section 7) in the built
library: see tools/isa_scan.py.  The shipped build must have none (tests/test_abi.py).

    python tools/exp/scan_pk_after_lds.py [max distance in instructions from the s_waitcnt; default: any]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
from isa_scan import scan  # noqa: E402
from vfml import hip  # noqa: E402

maxd = int(sys.argv[1]) if len(sys.argv) > 1 else None
total, counts = scan(hip.build(), maxd)
print(f"{counts['kernels']} functions, {counts['ds_read']} ds_read instructions, {counts['pk_f32']} packed-f32 ops")
print("packed-f32 ops that are the first reader of a ds_read result" + (f" within {maxd} instruction(s) of its s_waitcnt:" if maxd is not None else ":"))
for k, v in sorted(total.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(v):5d}  {k[:150]}")
    print("         e.g.", v[0][1])
print("kernels affected:", len(total))
