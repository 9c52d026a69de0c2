"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vfml.h declares.
No compute is launched here (no GPU in the CPU suite)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vfml.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vfml_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols():
    from vfml import hip
    path = hip.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vfml.h but not exported"
    assert sorted(hip.EXPORTS) == names
    lib.vfml_abi_version.restype = ctypes.c_int
    assert lib.vfml_abi_version() == 25


def test_argument_validation_needs_no_gpu():
    """Rejected descriptors return non-zero before any launch; error text is retrievable."""
    from vfml import hip
    L = hip.lib()
    d = hip.ConvDesc()
    assert L.vfml_conv2d(ctypes.byref(d), None) != 0
    assert b"null" in L.vfml_last_error()
    assert L.vfml_conv2d(None, None) != 0
    assert L.vfml_instnorm_workspace_bytes(2, 1024 * 3 + 1, 64) == 2 * 4 * 64 * 2 * 8
    assert L.vfml_corr_lookup(None, None, None, None, 4, 4, 1, 1, None, 4, None, 324, 0, 0, 0, None) != 0


def test_conv_desc_layout_matches_header():
    """ctypes mirror and C struct agree field-for-field (order and types)."""
    from vfml import hip
    text = open(os.path.join(ROOT, "include", "vfml.h")).read()
    body = re.search(r"typedef struct vfml_conv_desc \{(.*?)\} vfml_conv_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const float\* const\*|const float\*|const void\*|float\*|double\*|int32_t|float)\s+(.*)", decl)
        for name in m.group(2).split(","):
            fields.append((name.strip(), m.group(1)))
    want = {"const float* const*": ctypes.c_void_p, "const float*": ctypes.c_void_p, "const void*": ctypes.c_void_p, "float*": ctypes.c_void_p, "double*": ctypes.c_void_p, "int32_t": ctypes.c_int32,
            "float": ctypes.c_float}
    assert [(n, want[t]) for n, t in fields] == list(hip.ConvDesc._fields_)


def _gfx950_code_objects(lib_path):
    """Code objects of every translation unit in the library's .hip_fatbin section (clang offload bundles)."""
    import struct
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], blob.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + len(magic))
    return out


def test_no_kernel_spills_to_scratch():
    """Every kernel of the library keeps its state in registers: private (scratch) segment size 0 and no
    spilled VGPRs/SGPRs in the code-object metadata.  (hipcc silently moves accumulator arrays to scratch
    when it stops unrolling an MFMA loop - a 40x slowdown that no numerics test notices.)"""
    import re
    import subprocess
    import tempfile
    from vfml import hip
    hip.build()
    cos = _gfx950_code_objects(hip.LIB_PATH)
    assert len(cos) >= 5
    seen = 0
    for co in cos:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], check=True,
                                   capture_output=True, text=True).stdout
        for name, body in re.findall(r"\.name:\s+(\S+)(.*?)(?=\n\s+- \.|\Z)", notes, flags=re.S):
            vals = dict(re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", body))
            if "private_segment_fixed_size" not in vals:
                continue
            seen += 1
            assert int(vals["private_segment_fixed_size"]) == 0, f"{name} uses {vals['private_segment_fixed_size']} B of scratch"
            assert int(vals.get("vgpr_spill_count", 0)) == 0, f"{name} spills VGPRs"
    assert seen >= 30      # conv / GEMM instantiations + the streaming kernels


def test_no_packed_f32_first_reader_of_lds_results():
    """Round 2's wrong-result class, closed for the whole library: with one of this library's MFMA kernels on a second
    stream, a `v_pk_mul_f32` that was the FIRST reader of a `ds_read2_b32` result (straight behind the covering
    `s_waitcnt lgkmcnt(0)`) read the register's previous content in lanes 48-63 (profiles/r02_kernel_anatomy.md section 7).
    The engine runs two streams by default (encoder prefetch), so the shipped code object must not contain the pattern at
    all: no VALU op with a 64-bit register-pair operand - packed f32, and by the same operand form f64 / 64-bit integer ops
    - may be the first reader of a register a ds_read filled, at any distance from its wait.  Every source is built with
    -fno-slp-vectorize (vfml/hip.py COMMON_FLAGS) and the f64 norm-statistics folds read LDS through vfml_lds_f64."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from isa_scan import scan
    from vfml import hip
    assert "-fno-slp-vectorize" in hip.COMMON_FLAGS
    sites, counts = scan(hip.build(), maxd=None)
    assert counts["kernels"] >= 100 and counts["ds_read"] >= 3000, counts       # the scan saw the library
    assert not sites, "\n".join(f"{k}: {v[0][1]} ({len(v)} sites)" for k, v in sites.items())


def test_isa_scan_sees_the_pattern():
    """The scanner on the very sequence that failed (and on its repaired form)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from isa_scan import scan_disassembly
    bad = """
0000000000001000 <lookup>:
\tds_read2_b32 v[20:21], v11 offset0:10 offset1:11              // 000000001000: D86E0B0A 1400000B
\ts_waitcnt lgkmcnt(0)                                            // 000000001008: BF8CC07F
\tv_pk_mul_f32 v[16:17], v[16:17], v[20:21]                      // 00000000100C: D3B14010 18022910
"""
    good = bad.replace("v_pk_mul_f32 v[16:17], v[16:17], v[20:21]", "v_mul_f32_e32 v16, v16, v20")
    f64 = bad.replace("v_pk_mul_f32 v[16:17], v[16:17], v[20:21]", "v_add_f64 v[16:17], v[16:17], v[20:21]")
    moved = bad.replace("\tv_pk_mul_f32", "\tv_mov_b32_e32 v22, v20                    // 0: 0\n\tv_mov_b32_e32 v23, v21      // 0: 0\n\tv_pk_mul_f32")
    assert list(scan_disassembly(bad)[0]) == ["lookup"]
    assert list(scan_disassembly(f64)[0]) == ["lookup"]
    assert not scan_disassembly(good)[0]
    assert not scan_disassembly(moved)[0]
