"""ctypes binding of libvfml_hip.so (include/vfml.h).

PyTorch is used for device memory and streams only: every call takes raw device pointers
(`tensor.data_ptr()`) and launches on torch's current HIP stream.  There is NO CPU fallback:
if the library is missing or a kernel rejects its arguments this raises.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFML_LIB") or os.path.join(_HERE, "libvfml_hip.so")   # VFML_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["api.hip", "conv_gemm.hip", "conv_gemm_split.hip", "conv_gemm_tapx.hip", "stem.hip", "flow_half.hip", "enc_conv.hip", "norm_pool.hip", "flow_ops.hip", "effects.hip"]

STATS_ROWS_F32, STATS_ROWS_S16 = 128, 32    # pixels per stats_part block (include/vfml.h VFML_STATS_ROWS_*)
EPI_NONE, EPI_RELU, EPI_TANH, EPI_SIGMOID, EPI_TANH_RELU, EPI_GRU_ZR, EPI_GRU_Q, EPI_ADD_AUX = range(8)
FMT_F32, FMT_S16, FMT_F16 = 0, 1, 2     # storage formats (include/vfml.h; FMT_F16: correlation volumes only)


def vol_f16_levels(mask):
    """vol_fmt of a lookup over pyramids whose levels in `mask` (bit l = level l) hold f16 texels (VFML_VOL_F16_LEVELS)."""
    return 0x100 | mask


def _vol_mask(vol_fmt, levels):
    return (15 if vol_fmt == FMT_F16 else (vol_fmt & 15 if vol_fmt & 0x100 else 0)) & ((1 << levels) - 1)

KORDER_TAP, KORDER_CBLOCK, KORDER_CBLOCK64 = 0, 1, 2   # K-axis order of split weight planes (include/vfml.h)
CONV_SWAP_CROSS, CONV_MFMA2, CONV_MFMA1, CONV_MFMA2A, CONV_PER_TAP = 1, 2, 4, 8, 16   # vfml_conv_desc.flags


class ConvDesc(ctypes.Structure):
    """Mirror of `vfml_conv_desc` (include/vfml.h)."""
    _fields_ = [
        ("in0", c_void_p), ("c0", c_int32), ("ld0", c_int32),
        ("in1", c_void_p), ("c1", c_int32), ("ld1", c_int32),
        ("n", c_int32), ("h", c_int32), ("w", c_int32),
        ("weight", c_void_p), ("bias", c_void_p),
        ("cout", c_int32), ("kh", c_int32), ("kw", c_int32), ("stride", c_int32),
        ("pad_h", c_int32), ("pad_w", c_int32),
        ("out", c_void_p), ("ldo", c_int32),
        ("epilogue", c_int32), ("split", c_int32), ("out_scale", c_float),
        ("aux0", c_void_p), ("ld_aux0", c_int32),
        ("aux1", c_void_p), ("ld_aux1", c_int32),
        ("addend", c_void_p), ("ld_addend", c_int32),
        ("out_t", c_void_p), ("ld_out_t", c_int32),
        ("flags", c_int32),
        ("stats_part", c_void_p),
        ("ksplit_ws", c_void_p),
        ("proj_hi", c_void_p), ("proj_lo", c_void_p), ("proj_n", c_int32), ("proj_kp", c_int32), ("proj_scale", c_float),
        ("proj_out", c_void_p), ("ld_proj", c_int32),
        ("addend_ind", c_void_p),
    ]


# Every source is built without the SLP vectoriser.  Left on, it pairs adjacent scalar f32 multiplies / adds into
# v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32; where such a packed op was the FIRST reader of a ds_read result, straight behind
# the s_waitcnt that covers it, the correlation lookup read a stale register in lanes 48-63 whenever one of this library's
# MFMA kernels ran on another stream (profiles/r02_kernel_anatomy.md section 7) - and the engine runs two streams by default
# (the encoder prefetch).  Round 2 closed the one observed instance; round 3 closes the class: no packed-f32 first reader of
# an LDS result anywhere in the shipped code object (tests/test_abi.py::test_no_packed_f32_first_reader_of_lds_results).
# Packed f32 ops beside MFMAs are an anti-lever anyway (MI355X_MICROARCH.md, cycle constants: 2 v_pk_add_f32 per MFMA gap
# cost +26 cycles against 2 v_fma_f32).
COMMON_FLAGS = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {}


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into libvfml_hip.so (in-tree). Cross-compiles without a GPU."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, "vfml_common.h"), os.path.join(CSRC, "conv_split_common.h"),
            os.path.join(_HERE, "..", "..", "include", "vfml.h")]
    deps = srcs + hdrs
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    # one object per source, compiled side by side (objects kept under csrc/_obj: a changed source recompiles alone)
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + COMMON_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + [
                "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + [
        os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load the library (after torch, so that its libamdhip64.so.7 is the one HIP runtime in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vfml HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc). There is no CPU fallback for the flow engine.")
    L = ctypes.CDLL(LIB_PATH)
    L.vfml_last_error.restype = c_char_p
    L.vfml_abi_version.restype = c_int
    L.vfml_conv2d.argtypes = [POINTER(ConvDesc), c_void_p]
    L.vfml_conv2d_split.argtypes = [POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_int,
                                    c_void_p]
    L.vfml_to_s16.argtypes = [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_float, c_void_p]
    L.vfml_softmax_rows_s16.argtypes = [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int64, c_float, c_void_p]
    L.vfml_transpose_split_f16.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]
    L.vfml_split_f16.argtypes = [c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]
    L.vfml_frames_to_nhwc4.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p]
    L.vfml_instnorm_workspace_bytes.restype = c_int64
    L.vfml_instnorm_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.vfml_instnorm_stats.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]
    L.vfml_softmax_rows_f16.argtypes = [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int64, c_float, c_void_p]
    L.vfml_transpose_to_s16.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p]
    L.vfml_add_to_s16.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]
    L.vfml_instnorm_finalize.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int64, c_void_p]
    L.vfml_instnorm_finalize_workspace_bytes.restype = c_int64
    L.vfml_instnorm_finalize_workspace_bytes.argtypes = [c_int, c_int]
    L.vfml_stem7x7s2_chunks.argtypes = [c_int, c_int]
    L.vfml_stem7x7s2.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
    L.vfml_instnorm_apply.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]
    L.vfml_avgpool2x2.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.vfml_corr_lookup.argtypes = [POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                   c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    L.vfml_corr_lookup_indirect.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                            c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p]
    L.vfml_corr_lookup_indirect_bidir.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                                  c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                                  c_int, c_int, c_int, c_int, c_void_p]
    L.vfml_ptr_table_set.argtypes = [c_void_p, POINTER(c_void_p), c_int, c_void_p]
    L.vfml_tapsum3x3_update.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p,
                                        c_int, c_void_p, c_int, c_int, c_void_p]
    L.vfml_conv3x3_c64.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                   c_int, c_void_p, c_void_p]
    L.vfml_flow_half.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_float,
                                 c_void_p, c_void_p, c_int, c_void_p]
    L.vfml_coords_init.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p]
    L.vfml_flow_rows7.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    L.vfml_tapsum3x3.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, ctypes.c_int64, c_void_p]
    L.vfml_coords_update.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                                     c_int, c_void_p]
    L.vfml_flow_lod.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    L.vfml_flow_encode.argtypes = [c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p,
                                   c_void_p]
    L.vfml_taa_blend.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_double,
                                 c_double, c_void_p]
    L.vfml_flow_quality_map.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]
    L.vfml_convex_upsample.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = header/library drift
    if L.vfml_abi_version() != 25:
        raise RuntimeError("libvfml_hip.so ABI version mismatch")
    _lib = L
    return L


EXPORTS = [
    "vfml_conv2d", "vfml_conv2d_split", "vfml_split_f16", "vfml_to_s16", "vfml_softmax_rows_s16", "vfml_softmax_rows_f16", "vfml_transpose_to_s16", "vfml_add_to_s16",
    "vfml_transpose_split_f16", "vfml_frames_to_nhwc4", "vfml_instnorm_workspace_bytes", "vfml_instnorm_stats",
    "vfml_instnorm_apply", "vfml_instnorm_finalize", "vfml_instnorm_finalize_workspace_bytes", "vfml_avgpool2x2", "vfml_corr_lookup", "vfml_corr_lookup_indirect", "vfml_corr_lookup_indirect_bidir",
    "vfml_ptr_table_set", "vfml_coords_update", "vfml_coords_init", "vfml_tapsum3x3", "vfml_tapsum3x3_update", "vfml_flow_rows7", "vfml_flow_half", "vfml_conv3x3_c64",
    "vfml_convex_upsample", "vfml_stem7x7s2", "vfml_stem7x7s2_chunks", "vfml_flow_lod", "vfml_flow_encode", "vfml_taa_blend", "vfml_flow_quality_map", "vfml_last_error", "vfml_abi_version",
]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().vfml_last_error().decode()}")


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, offset=0):
    """Device pointer of a float32 tensor's storage start + `offset` floats (0/None-safe)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr() + 4 * offset)


def _dev(t, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"expected a contiguous {dtype} device tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t


# -- optional per-launch timing of vfml_conv2d (bench.py's roofline leg) ---------------------------
_PROFILE = None


_PROFILE_HBM = None


def profile_begin():
    """Start recording (kernel variant, algorithmic FLOPs, start/end HIP events) per vfml_conv2d launch, and
    (kernel, algorithmic bytes, events) per correlation lookup.
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""
    global _PROFILE, _PROFILE_HBM
    _PROFILE = []
    _PROFILE_HBM = []


def profile_end_hbm():
    """{kernel: {"launches", "bytes", "ms"}} of the HBM-bound launches recorded since profile_begin()
    (call before profile_end(), which stops the recording)."""
    rec = _PROFILE_HBM or []
    torch.cuda.synchronize()
    out = {}
    for name, nbytes, e0, e1 in rec:
        d = out.setdefault(name, {"launches": 0, "bytes": 0.0, "ms": 0.0})
        d["launches"] += 1
        d["bytes"] += nbytes
        d["ms"] += e0.elapsed_time(e1)
    return out


def profile_end():
    """Stop recording; returns {variant: {"launches", "flops", "bytes", "ms", "shapes"}} (synchronises); bytes = every operand
    read once and the result written once (the algorithmic HBM traffic of the launch); shapes = the same sums per layer
    shape "khxkw cin->cout" (one kernel variant serves several layers: bench.py's roofline.by_shape)."""
    global _PROFILE, _PROFILE_HBM
    rec, _PROFILE = _PROFILE or [], None
    _PROFILE_HBM = None
    torch.cuda.synchronize()
    out = {}
    for variant, flops, nbytes, e0, e1, shape in rec:
        d = out.setdefault(variant, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "shapes": {}})
        ms = e0.elapsed_time(e1)
        for t in (d, d["shapes"].setdefault(shape, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0})):
            t["launches"] += 1
            t["flops"] += flops
            t["bytes"] += nbytes
            t["ms"] += ms
    return out


def conv_variant(cout, split=False, ctot=32, in16=False, m=0, order=KORDER_TAP, plain_f32_out=False, fastk=False,
                 cswap=False, nm=3, same=None, per_tap=False, stats=False, h16=False):
    """Template instantiation vfml_conv2d[_split] dispatches to, spelled as rocprofv3 prints it (mirrors
    the dispatch at the end of csrc/conv_gemm_split.hip; the rare 64-wide choice of the register-staged
    split kernel for cout > 64 is not modelled)."""
    tile = "128, 2, 2" if cout > 64 else ("64, 2, 2" if cout > 32 else "32, 4, 1")
    if not split:
        return f"conv_gemm_kernel<{tile}>"
    dma = in16
    if dma:     # split-f16, LDS-DMA staged
        fk = "true" if fastk else "false"       # uniform-step loader (SplitArgs::fastk)
        if plain_f32_out and cout >= 1024 and cout % 4 == 0:
            if not fastk:
                nm = 3
            return f"conv_gemm_dma_kernel<2, 2, 2, 2, true, {fk}, {'true' if cswap else 'false'}, {nm}, false, {'true' if h16 else 'false'}>"   # persistent GEMM form
        if cout <= 32:
            return f"conv_gemm_dma_kernel<1, 1, 4, 1, false, false, false, {nm}, true, false>"
        # the shared-stage kernel (csrc/conv_gemm_tapx.hip: vfml_detail::tapx_cfg and the dispatcher's condition):
        # `same` = (kh, kw) of a stride-1 "same" convolution, else None
        tapx_mode = int(os.environ.get("VFML_TAPX", "1"))
        tapx = (tapx_mode and not per_tap and fastk and same is not None and 2 <= same[1] <= 5 and same[0] <= 4
                and nm in (3, 5))
        tiles256 = -(-m // 256)
        fills = tiles256 * 100 >= -(-tiles256 // 512) * 512 * 85      # 256-row tiles fill their last round to 85 %
        if tapx and cout <= 64 and fills:
            return f"conv_gemm_tapx_kernel<2, 2, 4, 1, {nm}>"
        if tapx and 64 < cout <= 96 and fills:
            return f"conv_gemm_tapx_kernel<2, 3, 4, 1, {nm}>"
        if cout <= 64:
            return f"conv_gemm_dma_kernel<2, 1, 2, 2, false, {fk}, false, {nm}, true, false>"
        def cost(tbm, tbn, mf, eff):
            tiles = -(-m // tbm) * -(-cout // tbn)
            return (tiles / 512.0 if tiles > 512 else 1.0) * mf / eff
        cands = [(cost(192, 128, 6.0, 1.0), "3, 2, 2, 2"), (cost(128, 192, 6.0, 1.0), "2, 3, 2, 2"),
                 (cost(128, 128, 4.0, 0.93), "2, 2, 2, 2"), (cost(128, 64, 2.0, 0.7), "2, 1, 2, 2")]
        t = cands[0][1]
        best = cands[0][0]
        for c, name in cands[1:]:
            if c < best:
                best, t = c, name
        if tapx and t in ("3, 2, 2, 2", "2, 3, 2, 2") and (nm == 5 or tapx_mode >= 2):
            return f"conv_gemm_tapx_kernel<{t}, {nm}>"
        return f"conv_gemm_dma_kernel<{t}, false, {fk}, false, {nm}, true, false>"
    bigc = "true" if (ctot >= 32 or in16) else "false"
    return f"conv_gemm_split_kernel<{tile}, {bigc}, {'true' if in16 else 'false'}, {nm}>"


class SplitWeight:
    """[rows][k] f32 matrix (times a power-of-two `scale`) as two f16 planes [rows][kp] (hi, lo) for
    vfml_conv2d_split."""

    def __init__(self, rows, k, device):
        self.rows, self.k, self.kp = rows, k, (k + 31) // 32 * 32
        self.scale = 1.0
        self.order = KORDER_TAP     # set to KORDER_CBLOCK by whoever fills it with pack_conv_weight(cblock=True)
        # one allocation: the LDS-DMA conv kernel reaches both planes through one buffer descriptor
        self.planes = torch.empty(2 * rows * self.kp, dtype=torch.float16, device=device)
        self.hi, self.lo = self.planes[:rows * self.kp], self.planes[rows * self.kp:]

    @staticmethod
    def auto_scale(absmax):
        """Largest power of two that keeps scale * absmax below 2^14 (f16 max is 65504)."""
        import math
        if not (absmax > 0.0) or not math.isfinite(absmax):
            return 1.0
        return 2.0 ** math.floor(math.log2(16384.0 / absmax))

    def fill_transposed(self, src, rows_src, ld=None, scale=1.0, src_off=0):
        """src: flat f32 [rows_src][self.rows] (row stride ld): planes of its transpose, k = source row."""
        self.scale = float(scale)
        _check(lib().vfml_transpose_split_f16(_ptr(_dev(src), src_off), rows_src, self.rows, ld or self.rows,
                                              self.scale, c_void_p(self.hi.data_ptr()), c_void_p(self.lo.data_ptr()),
                                              self.kp, _stream()), "vfml_transpose_split_f16")
        return self

    def fill(self, src, src_off=0, ld=None, scale=1.0):
        """src: flat f32 device tensor holding [rows][k] at float offset src_off with row stride ld."""
        self.scale = float(scale)
        _check(lib().vfml_split_f16(_ptr(_dev(src), src_off), self.rows, self.k, ld or self.k, self.scale,
                                    c_void_p(self.hi.data_ptr()), c_void_p(self.lo.data_ptr()), self.kp, _stream()),
               "vfml_split_f16")
        return self


def conv2d(in0, c0, ld0, n, h, w, weight, bias, cout, kh, kw, out, ldo, *, stride=1, pad_h=0, pad_w=0,
           in0_off=0, weight_off=0, in1=None, c1=0, ld1=0, in1_off=0, out_off=0, epilogue=EPI_NONE, split=0, out_scale=1.0,
           aux0=None, ld_aux0=0, aux0_off=0, aux1=None, ld_aux1=0, aux1_off=0,
           in_fmt=FMT_F32, out_fmt=FMT_F32, aux_fmt=FMT_F32, addend=None, ld_addend=0, addend_off=0,
           out_t=None, ld_out_t=0, out_t_off=0, swap_cross=False, stats_part=None, mfma=3, per_tap=False, ksplit_ws=None,
           proj=None, proj_out=None, ld_proj=0, addend_ind=None):
    """Launch vfml_conv2d. Tensors are flat float32 device buffers; *_off are float offsets into them
    (channel slices of wider NHWC buffers).  mfma: terms of the split-f16 product (3; 2 or "2w" = weights as plain
    f16; "2a" = activations as plain f16; 1 = both operands plain f16 - VFML_CONV_MFMA2 / _MFMA2A / _MFMA1).  per_tap:
    VFML_CONV_PER_TAP (the per-tap staging kernel where the shared-stage one would run; same bits).
    proj (a SplitWeight [proj_n <= 48][cout]) with proj_out / ld_proj: the projection epilogue (vfml_conv_desc.proj_out) -
    relu(out) is not stored, proj_out receives cout / 128 partial maps [n*ho*wo][ld_proj] of relu(out) x proj^T."""
    d = ConvDesc()
    d.in0, d.c0, d.ld0 = _ptr(_dev(in0), in0_off), c0, ld0
    d.in1, d.c1, d.ld1 = (_ptr(_dev(in1), in1_off) if in1 is not None else None), c1, ld1
    d.n, d.h, d.w = n, h, w
    is_split = isinstance(weight, (SplitWeight, PlainWeight))
    d.weight = None if is_split else _ptr(_dev(weight), weight_off)
    d.bias = _ptr(_dev(bias)) if bias is not None else None
    d.cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w = cout, kh, kw, stride, pad_h, pad_w
    d.out, d.ldo = _ptr(_dev(out), out_off), ldo
    d.epilogue, d.split, d.out_scale = epilogue, split, out_scale
    d.aux0, d.ld_aux0 = (_ptr(_dev(aux0), aux0_off) if aux0 is not None else None), ld_aux0
    d.aux1, d.ld_aux1 = (_ptr(_dev(aux1), aux1_off) if aux1 is not None else None), ld_aux1
    d.addend, d.ld_addend = (_ptr(_dev(addend), addend_off) if addend is not None else None), ld_addend
    d.out_t, d.ld_out_t = (_ptr(_dev(out_t), out_t_off) if out_t is not None else None), ld_out_t
    if mfma == "2w":
        mfma = 2
    if mfma not in (1, 2, 3, "2a") or (swap_cross and mfma != 3):
        raise ValueError(f"mfma={mfma!r}: 1, 2 ('2w'), '2a' or 3 (3 with swap_cross)")
    d.flags = (CONV_SWAP_CROSS if swap_cross else 0) | {3: 0, 2: CONV_MFMA2, "2a": CONV_MFMA2A, 1: CONV_MFMA1}[mfma] | (CONV_PER_TAP if per_tap else 0)
    d.stats_part = c_void_p(stats_part.data_ptr()) if stats_part is not None else None   # float64 workspace
    d.ksplit_ws = _ptr(_dev(ksplit_ws)) if ksplit_ws is not None else None      # GEMM form: second half of K (vfml.h)
    if addend_ind is not None:       # (table tensor, entry): the device cell that holds the addend pointer (vfml.h)
        tab, ent = addend_ind
        d.addend_ind = c_void_p(tab.data_ptr() + 8 * ent)
    if proj is not None:
        if not isinstance(proj, SplitWeight) or proj.lo is None or proj_out is None:
            raise ValueError("proj: a SplitWeight with both planes, and proj_out")
        d.proj_hi, d.proj_lo = c_void_p(proj.hi.data_ptr()), c_void_p(proj.lo.data_ptr())
        d.proj_n, d.proj_kp, d.proj_scale = proj.rows, proj.kp, proj.scale
        d.proj_out, d.ld_proj = _ptr(_dev(proj_out)), ld_proj
    if is_split:
        # weight_off counts rows of the split planes (each row kp halves)
        def launch():
            _check(lib().vfml_conv2d_split(ctypes.byref(d), c_void_p(weight.hi.data_ptr() + 2 * weight_off * weight.kp),
                                           c_void_p(weight.lo.data_ptr() + 2 * weight_off * weight.kp) if weight.lo is not None else None, weight.kp,
                                           weight.scale, in_fmt, out_fmt, aux_fmt, weight.order, _stream()),
                   "vfml_conv2d_split")
    else:
        if in_fmt != FMT_F32 or out_fmt != FMT_F32 or aux_fmt != FMT_F32:
            raise ValueError("the exact-f32 kernel (vfml_conv2d) takes and writes plain f32 activations only")

        def launch():
            _check(lib().vfml_conv2d(ctypes.byref(d), _stream()), "vfml_conv2d")
    if _PROFILE is None:
        launch()
        return
    ho = (h + 2 * pad_h - kh) // stride + 1
    wo = (w + 2 * pad_w - kw) // stride + 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    plain = (epilogue in (EPI_NONE, EPI_RELU) and addend is None and out_fmt in (FMT_F32, FMT_F16) and ldo % 4 == 0)
    ctot = c0 + c1
    pointwise = kh == 1 and kw == 1 and stride == 1 and pad_h == 0 and pad_w == 0
    fastk = (is_split and in_fmt == FMT_S16 and (weight.order in (KORDER_CBLOCK, KORDER_CBLOCK64) or (pointwise and ctot % 32 == 0))
             and c0 % 32 == 0 and ctot % 32 == 0 and kh * kw <= 32 and not os.environ.get("VFML_NO_FASTK")
             and (in1 is None or (ld1 == ld0 and in1.data_ptr() + 4 * in1_off >= in0.data_ptr() + 4 * in0_off)))
    nm_eff = {"2a": 4}.get(mfma, mfma)
    if (mfma == 1 and fastk and c0 % 64 == 0 and ctot % 64 == 0 and cout > 32 and
            (weight.order == KORDER_CBLOCK64 or pointwise) and not os.environ.get("VFML_NO_H64")):
        nm_eff = 5                                           # 64-channel steps of hi halves
    elif is_split and weight.lo is None:
        nm_eff = {3: 2, 2: 2, 4: 1, 1: 1}[nm_eff]            # a single weight plane has no lo half to use
    same = (kh, kw) if (stride == 1 and ho == h and wo == w and not pointwise) else None
    # (a VFML_FMT_F16 output is written by the GEMM form whatever its width)
    _PROFILE.append((conv_variant(max(cout, 1024) if out_fmt == FMT_F16 else cout, is_split, ctot, in_fmt == FMT_S16, n * ho * wo,
                                  weight.order if is_split else KORDER_TAP, plain, fastk, swap_cross, nm_eff, same, per_tap,
                                  h16=out_fmt == FMT_F16),
                     2.0 * n * ho * wo * (kh * kw * (c0 + c1) * cout + (cout * proj.rows if proj is not None else 0)),
                     # operands read once + result written once, 4 bytes per element in either activation format
                     4.0 * (n * h * w * (c0 + c1) + n * ho * wo * cout * (2 if out_t is not None else 1)
                            + cout * kh * kw * (c0 + c1) * (0.5 if is_split and weight.lo is None else 1.0)), e0, e1,
                     f"{kh}x{kw} {c0 + c1}->{cout}" + (f"->{proj.rows}" if proj is not None else "")))


def frames_to_nhwc4(src, n, H, W, scale, shift, dst):
    """src: uint8 [n,H,W,3] or float32 [n,3,H,W] device tensor -> dst float32 [n,H,W,4]."""
    if src.dtype == torch.uint8:
        kind = 0
    elif src.dtype == torch.float32:
        kind = 1
    else:
        raise ValueError(f"frames must be uint8 HWC or float32 CHW, got {src.dtype}")
    if not (src.is_cuda and src.is_contiguous()):
        raise ValueError("frames must be a contiguous device tensor")
    _check(lib().vfml_frames_to_nhwc4(c_void_p(src.data_ptr()), kind, n, H, W, scale, shift, _ptr(_dev(dst)),
                                      _stream()), "vfml_frames_to_nhwc4")


def instnorm_workspace_bytes(n, hw, c):
    return int(lib().vfml_instnorm_workspace_bytes(n, hw, c))


def instnorm_stats(x, n, hw, c, stats, workspace, eps=1e-5):
    _check(lib().vfml_instnorm_stats(_ptr(_dev(x)), n, hw, c, eps, _ptr(_dev(stats)),
                                     c_void_p(workspace.data_ptr()), _stream()), "vfml_instnorm_stats")


def instnorm_finalize_workspace_bytes(chunks, c):
    return int(lib().vfml_instnorm_finalize_workspace_bytes(chunks, c))


def instnorm_finalize(part, n, chunks, c, hw, stats, eps=1e-5, workspace=None):
    """Fold the partial sums a convolution left in `part` (conv2d(..., stats_part=part)) into {mean, rstd}.
    workspace: a float64 device tensor for the slice-wise first pass over many partials (instnorm_finalize_workspace_bytes;
    None: one is taken from torch's allocator for this call - stream-ordered, so concurrent streams never share it)."""
    need = instnorm_finalize_workspace_bytes(chunks, c)
    if need and (workspace is None or workspace.numel() * 8 < need):
        workspace = torch.empty(min(n, 8) * need // 8, dtype=torch.float64, device=stats.device)
    _check(lib().vfml_instnorm_finalize(c_void_p(part.data_ptr()), n, chunks, c, hw, eps, _ptr(_dev(stats)),
                                        c_void_p(workspace.data_ptr()) if workspace is not None else None,
                                        workspace.numel() * 8 if workspace is not None else 0, _stream()),
           "vfml_instnorm_finalize")


def stem_chunks(h, w):
    return int(lib().vfml_stem7x7s2_chunks(h, w))


def pack_stem_weight(w, device):
    """[64, 3|4, 7, 7] conv weight -> SplitWeight planes [64][224] in the stem kernel's K order (ky-major rows of 8 taps x 4
    channels, tap 7 and channel 3 zero)."""
    w = w.detach().to(device=device, dtype=torch.float32)
    cout, cin = w.shape[0], w.shape[1]
    if cout != 64 or cin not in (3, 4) or tuple(w.shape[2:]) != (7, 7):
        raise ValueError(f"pack_stem_weight: a [64, 3|4, 7, 7] weight expected, got {tuple(w.shape)}")
    k = torch.zeros(cout, 7, 8, 4, device=device)
    k[:, :, :7, :cin] = w.permute(0, 2, 3, 1)
    sw = SplitWeight(cout, 224, device)
    return sw.fill(k.reshape(-1).contiguous(), scale=SplitWeight.auto_scale(float(w.abs().max())))


def stem7x7s2(frames, n, h, w, weight, bias, out, stats_part=None):
    """vfml_stem7x7s2: frames f32 [n,h,w,4] -> out f32 [n,ho,wo,64] (+ per-tile statistics partials)."""
    if not isinstance(weight, SplitWeight) or weight.rows != 64 or weight.kp != 224:
        raise ValueError("stem7x7s2: weight must come from pack_stem_weight")
    _check(lib().vfml_stem7x7s2(_ptr(_dev(frames)), n, h, w, c_void_p(weight.hi.data_ptr()), c_void_p(weight.lo.data_ptr()),
                                weight.scale, _ptr(_dev(bias)) if bias is not None else None, _ptr(_dev(out)),
                                c_void_p(stats_part.data_ptr()) if stats_part is not None else None, _stream()), "vfml_stem7x7s2")


def instnorm_apply(x, stats, n, hw, c, out, res=None, res_stats=None, out_fmt=FMT_F32):
    """out_fmt FMT_S16: `out` (and a `res` without res_stats, an earlier out) are split rows."""
    _check(lib().vfml_instnorm_apply(_ptr(_dev(x)), _ptr(_dev(stats)), _ptr(res), _ptr(res_stats), n, hw, c,
                                     _ptr(_dev(out)), out_fmt, _stream()), "vfml_instnorm_apply")


def avgpool2x2(x, n, h, w, c, out):
    _check(lib().vfml_avgpool2x2(_ptr(_dev(x)), n, h, w, c, _ptr(_dev(out)), _stream()), "vfml_avgpool2x2")


def softmax_rows_s16(x, rows, cols, ld_in, out, ld_out, x_off=0, out_off=0, scale=1.0):
    """scale * row softmax of f32 scores -> split rows (FMT_S16), zero-filled to ld_out."""
    _check(lib().vfml_softmax_rows_s16(_ptr(_dev(x), x_off), rows, cols, ld_in, _ptr(_dev(out), out_off), ld_out,
                                       float(scale), _stream()), "vfml_softmax_rows_s16")


class PlainWeight:
    """One f16 plane [rows][kp] as the second operand of a GEMM (vfml_conv2d_split with w_lo == NULL)."""
    lo = None
    order = KORDER_TAP

    def __init__(self, rows, kp, device, scale=1.0):
        if kp % 32:
            raise ValueError("PlainWeight: kp must be a multiple of 32")
        self.rows, self.k, self.kp, self.scale = rows, kp, kp, float(scale)
        self.hi = torch.empty(rows * kp, dtype=torch.float16, device=device)


def softmax_rows_f16(x, rows, cols, ld_in, weight, x_off=0):
    """weight.scale * row softmax of f32 scores -> the f16 plane of a PlainWeight (rows of weight.kp halves)."""
    _check(lib().vfml_softmax_rows_f16(_ptr(_dev(x), x_off), rows, cols, ld_in, c_void_p(weight.hi.data_ptr()), weight.kp,
                                       weight.scale, _stream()), "vfml_softmax_rows_f16")


def transpose_to_s16(src, rows, c, ld, dst, ld_dst, scale=1.0, src_off=0, dst_off=0):
    """f32 [rows][c] -> split rows [c][ld_dst] of its transpose times scale."""
    _check(lib().vfml_transpose_to_s16(_ptr(_dev(src), src_off), rows, c, ld, float(scale), _ptr(_dev(dst), dst_off), ld_dst,
                                       _stream()), "vfml_transpose_to_s16")


def add_to_s16(x, ldx, aux, ld_aux, out, ld_out, rows, c, scale=1.0, x_off=0, aux_off=0, out_off=0):
    """out = aux + scale * x (x f32, aux / out split rows)."""
    _check(lib().vfml_add_to_s16(_ptr(_dev(x), x_off), ldx, _ptr(_dev(aux), aux_off), ld_aux, _ptr(_dev(out), out_off), ld_out,
                                 rows, c, float(scale), _stream()), "vfml_add_to_s16")


def to_s16(src, rows, c, ld_src, dst, ld_dst, src_off=0, dst_off=0, scale=1.0):
    """scale * f32 rows [rows][c] -> split rows (FMT_S16) at float offset dst_off of dst."""
    _check(lib().vfml_to_s16(_ptr(_dev(src), src_off), rows, c, ld_src, _ptr(_dev(dst), dst_off), ld_dst, float(scale),
                             _stream()), "vfml_to_s16")


def ptr_table_set(table, tensors):
    """Write the device pointers of `tensors` into `table` (an int64 device tensor), asynchronously on the current stream."""
    n = len(tensors)
    if not (table.is_cuda and table.dtype == torch.int64 and table.numel() >= n):
        raise ValueError("ptr_table_set: table must be an int64 device tensor with room for the pointers")
    ptrs = (c_void_p * n)(*[t.data_ptr() for t in tensors])
    _check(lib().vfml_ptr_table_set(c_void_p(table.data_ptr()), ptrs, n, _stream()), "vfml_ptr_table_set")


def corr_lookup(pyrs, hl, wl, ld, radius, q_per_map, coords, coords_off, ld_coords, out, out_off, ld_out,
                out_fmt=FMT_F32, table=None, nmaps=None, vol_fmt=FMT_F32, vol_tile=0, bidir=None):
    """pyrs: list (one entry per query map) of lists (one flat float32 device tensor per level, rows =
    that map's q_per_map queries).  Queries / coords / out rows are ordered map-major.
    table (with nmaps): instead of `pyrs`, an int64 device tensor holding the same pointers, map-major
    (ptr_table_set), read when the kernel runs (vfml_corr_lookup_indirect).
    vol_tile: 0 (row-major level images, rows in query order) or VolTile.code (include/vfml.h).
    bidir = (dir_coords, dir_out, dir_tab) with a table: both directions of the nmaps query maps in one launch
    (vfml_corr_lookup_indirect_bidir)."""
    L = len(hl)
    if bidir is not None:
        dir_coords, dir_out, dir_tab = bidir

        def launch():
            _check(lib().vfml_corr_lookup_indirect_bidir(c_void_p(table.data_ptr()), (c_int32 * L)(*hl), (c_int32 * L)(*wl),
                                                         (c_int32 * L)(*ld), L, radius, nmaps, q_per_map,
                                                         _ptr(_dev(coords), coords_off), ld_coords, dir_coords,
                                                         _ptr(_dev(out), out_off), ld_out, dir_out, dir_tab, out_fmt, vol_fmt,
                                                         vol_tile, _stream()), "vfml_corr_lookup_indirect_bidir")
    elif table is not None:
        def launch():
            _check(lib().vfml_corr_lookup_indirect(c_void_p(table.data_ptr()), (c_int32 * L)(*hl), (c_int32 * L)(*wl),
                                                   (c_int32 * L)(*ld), L, radius, nmaps, q_per_map,
                                                   _ptr(_dev(coords), coords_off), ld_coords, _ptr(_dev(out), out_off),
                                                   ld_out, out_fmt, vol_fmt, vol_tile, _stream()), "vfml_corr_lookup_indirect")
    else:
        if pyrs and torch.is_tensor(pyrs[0]):
            pyrs = [pyrs]
        nmaps, L = len(pyrs), len(pyrs[0])
        ptrs = (c_void_p * (nmaps * L))(*[p.data_ptr() for m in pyrs for p in m])

        def launch():
            _check(lib().vfml_corr_lookup(ptrs, (c_int32 * L)(*hl), (c_int32 * L)(*wl), (c_int32 * L)(*ld), L, radius,
                                          nmaps, q_per_map, _ptr(_dev(coords), coords_off), ld_coords,
                                          _ptr(_dev(out), out_off), ld_out, out_fmt, vol_fmt, vol_tile, _stream()), "vfml_corr_lookup")
    if _PROFILE_HBM is None:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    # algorithmic bytes: per query and level the (2r+2)^2 integer-grid patch in, (2r+1)^2 samples out
    q = (2 if bidir is not None else 1) * nmaps * q_per_map
    m16 = _vol_mask(vol_fmt, L)
    texel_bytes = sum(2.0 if (m16 >> l) & 1 else 4.0 for l in range(L))
    _PROFILE_HBM.append(("corr_lookup", q * ((2 * radius + 2) ** 2 * texel_bytes + L * (2 * radius + 1) ** 2 * 4.0), e0, e1))


class VolTile:
    """Tiled layout of the correlation volume (include/vfml.h vfml_corr_lookup, vol_tile): level images - and the level-0
    grid that orders the volume's rows - stored as 2^tws x 2^ths tiles.  The volume GEMM produces it for free when the rows
    of both its operands are in tile order: `position(h, w)` is where each row-major pixel goes, `count(h, w)` how many rows
    the whole tiles take (positions no pixel maps to are zero rows: their volume entries are never read)."""

    def __init__(self, tws, ths):
        self.tws, self.ths = tws, ths
        self.code = tws + 16 * ths
        self._pos = {}

    def count(self, h, w):
        return ((((h - 1) >> self.ths) + 1) * (((w - 1) >> self.tws) + 1)) << (self.tws + self.ths)

    def position(self, h, w, dev):
        key = (h, w, str(dev))
        if key not in self._pos:
            y = torch.arange(h, device=dev, dtype=torch.int64).view(h, 1)
            x = torch.arange(w, device=dev, dtype=torch.int64).view(1, w)
            tpr = ((w - 1) >> self.tws) + 1
            pos = ((((y >> self.ths) * tpr + (x >> self.tws)) << (self.tws + self.ths))
                   + ((y & ((1 << self.ths) - 1)) << self.tws) + (x & ((1 << self.tws) - 1)))
            self._pos[key] = pos.reshape(-1).contiguous()
        return self._pos[key]

    def rows(self, x, h, w, c):
        """x: flat [h*w*c] row-major pixels -> flat [count(h, w)*c] in tile order, zero rows where no pixel lands."""
        n = self.count(h, w)
        out = torch.zeros(n, c, device=x.device, dtype=x.dtype)
        out.index_copy_(0, self.position(h, w, x.device), x.view(h * w, c))
        return out.view(-1)


def coords_init(coords1, n, h, w):
    _check(lib().vfml_coords_init(_ptr(_dev(coords1)), n, h, w, _stream()), "vfml_coords_init")


def flow_rows7(flow, n, h, w, rows):
    """rows[p] = the seven horizontal taps' flow quads of pixel p as 32 split-row channels (include/vfml.h vfml_flow_rows7)."""
    _check(lib().vfml_flow_rows7(_ptr(_dev(flow)), n, h, w, _ptr(_dev(rows)), _stream()), "vfml_flow_rows7")


def tapsum3x3(t, ld_t, bias, n, h, w, out, parts=1, part_stride=0):
    """out[p][0:4] = bias + the nine taps' quads of the tap-major 36-column map t (include/vfml.h vfml_tapsum3x3); parts > 1: t is
    that many maps part_stride floats apart whose sum is meant (conv2d(..., proj_out=))."""
    _check(lib().vfml_tapsum3x3(_ptr(_dev(t)), ld_t, _ptr(bias) if bias is not None else None, n, h, w, _ptr(_dev(out)),
                                parts, part_stride, _stream()), "vfml_tapsum3x3")


def conv3x3_c64(src, ld_in, n, h, w, weight, bias, out, ldo, stats_part=None, src_off=0):
    """The encoders' 64 -> 64 channel 3x3 convolution over split rows with its norm partial sums (include/vfml.h
    vfml_conv3x3_c64; image width a multiple of 32, weights in KORDER_CBLOCK order)."""
    _check(lib().vfml_conv3x3_c64(_ptr(_dev(src), src_off), ld_in, n, h, w, c_void_p(weight.hi.data_ptr()),
                                  c_void_p(weight.lo.data_ptr()), weight.kp, weight.scale,
                                  _ptr(_dev(bias)) if bias is not None else None, _ptr(_dev(out)), ldo,
                                  c_void_p(stats_part.data_ptr()) if stats_part is not None else None, _stream()),
           "vfml_conv3x3_c64")


def flow_half(flow, n, h, w, w1, b1, w2, b2, out, ld_out, out_off=0):
    """relu(convf2(relu(convf1(flow)))) of the motion encoder as one launch (include/vfml.h vfml_flow_half): w1 / w2 the
    layers' SplitWeights (rows7 layout / 64-channel-block order), plain f16 products."""
    _check(lib().vfml_flow_half(_ptr(_dev(flow)), n, h, w, c_void_p(w1.hi.data_ptr()), w1.kp, w1.scale, _ptr(_dev(b1)),
                                c_void_p(w2.hi.data_ptr()), w2.kp, w2.scale, _ptr(_dev(b2)), _ptr(_dev(out), out_off), ld_out,
                                _stream()), "vfml_flow_half")


def tapsum3x3_update(t, ld_t, bias, n, h, w, coords1, parts=1, part_stride=0, flow_a=None, ld_a=0, flow_a_off=0, flow_b=None,
                     ld_b=0, flow_b_off=0, fmt_b=FMT_F32):
    """tapsum3x3 and coords_update(delta = that sum) as one launch (include/vfml.h vfml_tapsum3x3_update)."""
    _check(lib().vfml_tapsum3x3_update(_ptr(_dev(t)), ld_t, _ptr(bias) if bias is not None else None, n, h, w, parts, part_stride,
                                       _ptr(_dev(coords1)), _ptr(flow_a, flow_a_off), ld_a, _ptr(flow_b, flow_b_off), ld_b,
                                       fmt_b, _stream()), "vfml_tapsum3x3_update")


def coords_update(coords1, delta, n, h, w, flow_a=None, ld_a=0, flow_a_off=0, flow_b=None, ld_b=0, flow_b_off=0,
                  fmt_b=FMT_F32):
    _check(lib().vfml_coords_update(_ptr(_dev(coords1)), _ptr(delta), n, h, w,
                                    _ptr(flow_a, flow_a_off), ld_a, _ptr(flow_b, flow_b_off), ld_b, fmt_b, _stream()),
           "vfml_coords_update")


def flow_lods(flow, num_lods=5):
    """[H,W,2] float32 device tensor -> list of `num_lods` device tensors (level 0 is `flow` itself)."""
    lods = [_dev(flow.contiguous())]
    for _ in range(1, num_lods):
        h, w = lods[-1].shape[:2]
        out = torch.empty((h + 1) // 2, (w + 1) // 2, 2, device=flow.device, dtype=torch.float32)
        _check(lib().vfml_flow_lod(_ptr(lods[-1]), h, w, _ptr(out), _stream()), "vfml_flow_lod")
        lods.append(out)
    return lods


ENCODE_GAMEDEV, ENCODE_RG8, ENCODE_RGB8 = 0, 1, 2


def flow_encode(flow, mode, clamp_range, width=1.0, height=1.0, scale=1.0):
    """[H,W,2] float32 device tensor -> [H,W,3] uint8 device tensor (vfml_flow_encode)."""
    import numpy as np
    f = _dev(flow.contiguous())
    h, w = f.shape[:2]
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=f.device)
    _check(lib().vfml_flow_encode(_ptr(f), h, w, mode, float(np.float32(width)), float(np.float32(height)),
                                  float(np.float32(scale)), float(np.float32(clamp_range)),
                                  float(np.float32(2 * clamp_range)), c_void_p(out.data_ptr()), _stream()),
           "vfml_flow_encode")
    return out


TAA_SIMPLE, TAA_BILINEAR, TAA_BILATERAL = 0, 1, 2
_PIX = {torch.uint8: 0, torch.float32: 1, torch.float64: 2}


def taa_blend(current, flow, history, mode, alpha, sigma_color=25.0):
    """One TAA step on the device (vfml_taa_blend): current [H,W,3] u8/f32, flow [H,W,2] f32 or None, history
    [H,W,3] f32/f64 -> new history, in the dtype the reference's numpy arithmetic gives it."""
    cur, hist = current.contiguous(), history.contiguous()
    if not (cur.is_cuda and hist.is_cuda and cur.dtype in (torch.uint8, torch.float32)
            and hist.dtype in (torch.float32, torch.float64)):
        raise ValueError(f"taa_blend: device tensors u8/f32 + f32/f64 expected, got {cur.dtype} {hist.dtype}")
    h, w = cur.shape[:2]
    if tuple(cur.shape) != (h, w, 3) or tuple(hist.shape) != (h, w, 3):
        raise ValueError(f"taa_blend: [H,W,3] images expected, got {tuple(cur.shape)} {tuple(hist.shape)}")
    if mode != TAA_SIMPLE:
        flow = _dev(flow.contiguous())
        if tuple(flow.shape) != (h, w, 2):
            raise ValueError(f"taa_blend: flow {tuple(flow.shape)} does not match the frame {h}x{w}")
    out_dtype = hist.dtype if mode == TAA_SIMPLE else (torch.float32 if mode == TAA_BILINEAR else torch.float64)
    out = torch.empty((h, w, 3), dtype=out_dtype, device=cur.device)
    _check(lib().vfml_taa_blend(c_void_p(cur.data_ptr()), _PIX[cur.dtype], None if mode == TAA_SIMPLE else _ptr(flow),
                                c_void_p(hist.data_ptr()), _PIX[hist.dtype], c_void_p(out.data_ptr()), _PIX[out_dtype],
                                h, w, mode, float(alpha), float(sigma_color), _stream()), "vfml_taa_blend")
    return out


def flow_quality_map(frame1, frame2, flow, threshold):
    """uint8 frames [H,W,3] + flow [fh,fw,2] f32 (device tensors) -> uint8 quality map [H,W,3] (vfml_flow_quality_map)."""
    f1, f2, fl = _dev(frame1.contiguous(), torch.uint8), _dev(frame2.contiguous(), torch.uint8), _dev(flow.contiguous())
    h, w = f1.shape[:2]
    if tuple(f1.shape) != (h, w, 3) or f2.shape != f1.shape or fl.dim() != 3 or fl.shape[2] != 2:
        raise ValueError(f"flow_quality_map: frames {tuple(f1.shape)} {tuple(f2.shape)}, flow {tuple(fl.shape)}")
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=f1.device)
    _check(lib().vfml_flow_quality_map(c_void_p(f1.data_ptr()), c_void_p(f2.data_ptr()), _ptr(fl), fl.shape[0], fl.shape[1],
                                       h, w, float(threshold), c_void_p(out.data_ptr()), _stream()), "vfml_flow_quality_map")
    return out


def convex_upsample(coords1, coords_off, ch, mask, mask_off, ld_mask, h, w, out, out_off=0):
    _check(lib().vfml_convex_upsample(_ptr(_dev(coords1), coords_off), ch, _ptr(_dev(mask), mask_off), ld_mask, h, w,
                                      _ptr(_dev(out), out_off), _stream()), "vfml_convex_upsample")
