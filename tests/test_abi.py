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
    assert lib.vfml_abi_version() == 7


def test_argument_validation_needs_no_gpu():
    """Rejected descriptors return non-zero before any launch; error text is retrievable."""
    from vfml import hip
    L = hip.lib()
    d = hip.ConvDesc()
    assert L.vfml_conv2d(ctypes.byref(d), None) != 0
    assert b"null" in L.vfml_last_error()
    assert L.vfml_conv2d(None, None) != 0
    assert L.vfml_instnorm_workspace_bytes(2, 1024 * 3 + 1, 64) == 2 * 4 * 64 * 2 * 8
    assert L.vfml_corr_lookup(None, None, None, None, 4, 4, 1, 1, None, 4, None, 324, 0, None) != 0


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
        m = re.match(r"(const float\*|float\*|int32_t|float)\s+(.*)", decl)
        for name in m.group(2).split(","):
            fields.append((name.strip(), m.group(1)))
    want = {"const float*": ctypes.c_void_p, "float*": ctypes.c_void_p, "int32_t": ctypes.c_int32,
            "float": ctypes.c_float}
    assert [(n, want[t]) for n, t in fields] == list(hip.ConvDesc._fields_)
