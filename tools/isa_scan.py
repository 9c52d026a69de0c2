"""Static scan of the library's gfx950 code for the pattern behind the two-stream lookup finding
(profiles/r02_kernel_anatomy.md section 7): a VALU op with a 64-bit register-pair operand - packed f32 (v_pk_*_f32: the observed failure), f64 and 64-bit integer
ops - that is the FIRST reader of a register filled by a ds_read.  `scan(lib_path, maxd)` -> {kernel: [(op, instruction text, instructions since the covering
s_waitcnt lgkmcnt)]}; maxd bounds that distance (None: any distance).  Used by tests/test_abi.py (must be empty for the
shipped library) and tools/exp/scan_pk_after_lds.py (prints the sites)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []


def scan_disassembly(dis, maxd=None):
    total, counts = {}, {"kernels": 0, "ds_read": 0, "pk_f32": 0}
    kern, fresh, since_wait = None, {}, 10 ** 9
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            kern, fresh, since_wait = m.group(1), {}, 10 ** 9
            counts["kernels"] += 1
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
        dst = _regs(args[0]) if args else []
        srcs = [r for a in args[1:] for r in _regs(a.split(" ")[0])]
        if op.startswith("ds_read"):
            counts["ds_read"] += 1
            for r in dst:
                fresh[r] = True
            since_wait += 1
            continue
        if op.startswith("ds_write") or op.startswith("global_store") or op.startswith("buffer_store"):
            srcs = [r for a in args for r in _regs(a.split(" ")[0])]
            dst = []
        # 64-bit-operand VALU ops: packed f32 (the observed failure) and, by the same operand form, f64 / 64-bit integer ops
        packed = re.match(r"v_pk_\w+_f32|v_\w+_f64|v_\w+_[biu]64", op) is not None
        counts["pk_f32"] += packed
        hit = [r for r in srcs if fresh.get(r)]
        if hit and packed and (maxd is None or since_wait <= maxd):
            total.setdefault(kern, []).append((op, line.split("//")[0].strip(), since_wait))
        for r in srcs:
            fresh.pop(r, None)
        for r in dst:
            fresh.pop(r, None)
        since_wait += 1
    return total, counts


def scan(lib_path, maxd=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_abi import _gfx950_code_objects
    total, counts = {}, {"kernels": 0, "ds_read": 0, "pk_f32": 0}
    with tempfile.TemporaryDirectory() as tmp:
        for n, co in enumerate(_gfx950_code_objects(lib_path)):
            path = os.path.join(tmp, f"{n}.co")
            with open(path, "wb") as f:
                f.write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", path], check=True, capture_output=True, text=True).stdout
            t, c = scan_disassembly(dis, maxd)
            total.update(t)
            for k in counts:
                counts[k] += c[k]
    return total, counts
