"""Static scan of the library's gfx950 code for the pattern behind the two-stream lookup finding
(profiles/r02_kernel_anatomy.md section 7): a packed-f32 VALU op (v_pk_mul/add/fma_f32) that is the FIRST reader of a
register filled by a ds_read, issued right behind the s_waitcnt that covers it.

    python tools/exp/scan_pk_after_lds.py [max distance in instructions from the s_waitcnt, default 2]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "video-flow-ml_amd"))
from test_abi import _gfx950_code_objects
from vfml import hip
maxd = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []


total = {}
with tempfile.TemporaryDirectory() as tmp:
    for n, co in enumerate(_gfx950_code_objects(hip.build())):
        path = os.path.join(tmp, f"{n}.co")
        open(path, "wb").write(co)
        dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", path], check=True, capture_output=True, text=True).stdout
        kern, fresh, since_wait = None, {}, 99
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                kern, fresh, since_wait = m.group(1), {}, 99
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
            if not m or kern is None:
                continue
            op, args = m.group(1), [a.strip() for a in m.group(2).split(",")] if m.group(2) else []
            if op == "s_waitcnt" and "lgkmcnt" in line:
                since_wait = 0
                continue
            if op.startswith("s_nop"):
                since_wait += 1
                continue
            dst = regs(args[0]) if args else []
            srcs = [r for a in args[1:] for r in regs(a.split(" ")[0])]
            if op.startswith("ds_read"):
                for r in dst:
                    fresh[r] = True
                since_wait += 1
                continue
            if op.startswith("ds_write") or op.startswith("global_store") or op.startswith("buffer_store"):
                srcs = [r for a in args for r in regs(a.split(" ")[0])]
                dst = []
            hit = [r for r in srcs if fresh.get(r)]
            if hit and re.match(r"v_pk_(mul|add|fma)_f32", op) and since_wait <= maxd:
                total.setdefault(kern, []).append((op, line.split("//")[0].strip(), since_wait))
            for r in srcs:
                fresh.pop(r, None)
            for r in dst:
                fresh.pop(r, None)
            since_wait += 1
print(f"packed-f32 ops that are the first reader of a ds_read result within {maxd} instruction(s) of its s_waitcnt:")
for k, v in sorted(total.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(v):5d}  {k[:150]}")
    print("         e.g.", v[0][1])
print("kernels affected:", len(total))
