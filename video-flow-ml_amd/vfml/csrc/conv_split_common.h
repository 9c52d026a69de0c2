// Shared by the split-f16 convolution translation units (conv_gemm_split.hip, conv_gemm_tapx.hip): argument block,
// compile-time loops, the LDS-DMA primitive, the bank swizzle of the 16x16x32 LDS image and the epilogue that turns a
// workgroup's fp32 tile in LDS into global rows.
#pragma once
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <utility>
#include "vfml_common.h"


namespace vfml_detail {
struct SplitArgs {
  const float* in0; const float* in1;
  const _Float16* whi; const _Float16* wlo; const float* bias;
  const float* aux0; const float* aux1; const float* addend;
  float* out;
  int c0, ld0, c1, ld1, ctot, ld_addend;
  int H, W, ho, wo;
  int kh, kw, stride, pad_h, pad_w;
  int M, K, Kp, cout;
  int d0off, d1off;             // float offsets of in0 / in1 from the common descriptor base (in0 field)
  int bytes0, bytesw;           // descriptor extents: sources (both, from the base) and weight planes
  int ldo, ld_aux0, ld_aux1;
  int epilogue, split;
  float out_scale, w_inv;
  int mtiles, ntiles;
  int vec_ok;  // out/aux/bias 16-byte aligned and ldo, ld_aux % 4 == 0 -> float4 epilogue
  int out16, aux16;   // output / aux operands in the split-row format (VFML_FMT_S16)
  int out_h16;        // GEMM form: out / out_t hold one f16 per element (VFML_FMT_F16)
  // LDS-DMA kernel: both weight planes through one descriptor at wbase (byte offsets of the planes, extent)
  const char* wbase; int whi_off, wlo_off, bytesb;
  int korder;   // VFML_KORDER_*
  int direct;   // LDS-DMA kernel: plain f32 output written straight from the accumulators
  int pointwise;  // 1x1 / stride 1 / no padding
  // uniform-step loader of the LDS-DMA kernel (channel-block order, whole 32-channel blocks per source,
  // one row stride): per K step the tap / channel offset is one scalar
  int fastk, abias, src1_delta;
  int tilebase;   // 1x1 over one source: the source descriptor starts at the tile's first row (sources > 2 GiB)
  float* out_t; int ld_out_t;   // GEMM form: transposed second output (or null)
  int cswap;                    // VFML_CONV_SWAP_CROSS
  int bhi;                      // LDS-DMA GEMM form: the weight operand is one plain f16 plane (no lo plane)
  double* stats_part;           // register-staged kernel: per row tile and channel {sum, sum of squares} of the result
  int nm;                       // terms of the split product: 3 all, 2 weights as plain f16, 4 activations as plain f16, 1 both
  int fast_epi;                 // 0: the general epilogue routine everywhere (VFML_FAST_EPI=0, A/B)
  int ksplit; float* out_k1; float* out_t_k1;   // persistent GEMM form: 2 = two work items per tile, one per half of K; the second half's sums go to out_k1 (out_t_k1)
  // projection epilogue (vfml_conv_desc.proj_*): relu(out) is not stored but multiplied, per 128-column tile, by that
  // tile's slice of a second [proj_n][cout] weight (two f16 planes, lo plane proj_lo_off bytes behind the hi plane)
  const char* proj_w; int proj_lo_off, proj_bytes, proj_n, proj_kp; float proj_inv; float* proj_out; int ld_proj;
  const float* const* addend_ind;   // vfml_conv_desc.addend_ind: a device cell holding the addend pointer to use (or null)
};
// conv_gemm_tapx.hip: the kernel that shares one activation stage between the taps of a filter row
#define VFML_TAPX_KWMAX 5      // widest filter row it is built for
int tapx_cfg(const SplitArgs& a, int cfg, bool forced);       // cfg = tile shape as TM TN WM WN digits; 0: not its call
int launch_tapx(SplitArgs& a, int cfg, hipStream_t s);
}  // namespace vfml_detail
using vfml_detail::SplitArgs;

namespace {

// loops over compile-time indices that cannot be left to the unroller (past its size budget hipcc keeps the loop and
// the accumulator arrays it indexes go to scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int KG = BK / 8;  // 16-byte units (8 halves) per row per K step

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 16-byte load through a buffer descriptor: an offset beyond num_records returns zeros, which is how
// padding taps, K tails and out-of-range rows/columns are filled (no branch, no select on the data).
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, 0, 0));
}
constexpr int OOB = 0x7fffffff;   // >= any num_records we create (all < 2^31 bytes)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// the addend of a call: vfml_conv_desc.addend, or what the device cell addend_ind holds when the kernel runs (a launch in a
// replayed HIP graph whose per-pixel bias lives somewhere else from field to field).  Explicitly GLOBAL: a pointer that
// went through a select or through memory would be dereferenced with flat loads.
typedef const __attribute__((address_space(1))) float* vfml_gfptr;
typedef const __attribute__((address_space(1))) f32x4* vfml_gf4ptr;
__device__ __forceinline__ vfml_gfptr addend_of(const SplitArgs& a) {
  return (vfml_gfptr)(a.addend_ind ? *a.addend_ind : a.addend);
}

union U8 {
  h16x8 v;
  h16x2 p[4];
};

// x (4 floats) -> hi/lo halves at element offset `at` (0 or 4) of the 8-wide units
__device__ __forceinline__ void split4(const f32x4 x, U8& hi, U8& lo, int at) {
#pragma unroll
  for (int e = 0; e < 2; ++e) vfml_split2(x[2 * e], x[2 * e + 1], hi.p[at / 2 + e], lo.p[at / 2 + e]);
}

__device__ __forceinline__ float epi1(float v, int epilogue, bool lowhalf, float x0, float x1) {
  switch (epilogue) {
    case VFML_EPI_RELU: return fmaxf(v, 0.f);
    case VFML_EPI_TANH: return tanhf(v);
    case VFML_EPI_SIGMOID: return sigmoidf_(v);
    case VFML_EPI_TANH_RELU: return lowhalf ? tanhf(v) : fmaxf(v, 0.f);
    case VFML_EPI_GRU_ZR: v = sigmoidf_(v); return lowhalf ? v : v * x0;
    case VFML_EPI_GRU_Q: return (1.f - x0) * x1 + x0 * tanhf(v);
    case VFML_EPI_ADD_AUX: return x0 + v;
    default: return v;
  }
}

// accumulator tiles of one wave -> the workgroup's fp32 tile in LDS (row stride LDC floats)
template <int TM, int TN, int LDC>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[TM][TN], float* sC, int row0, int col0, int r, int half) {
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const int col = col0 + j * 32 + r;
        sC[row * LDC + col] = acc[i][j][e];
      }
}

// the LDS tile -> global rows: bias, addend, activation / GRU gate math, f32 or split-row stores
template <int BN, int NT>
__device__ __forceinline__ void epilogue_rows(const SplitArgs& a, const float* sC, int m0, int n0, int t,
                                              int nrows = BM, int rstride = 32, int roff = 0) {
  // LDS row `row` is output pixel m0 + (row / 32) * rstride + roff + row % 32 (identity by default; the
  // LDS-DMA kernel passes the tile through in slabs of one 32-row block per wave row)
  constexpr int LDC = BN + 4;
  // 8 channels (one split-row unit) per thread, as two quads
  constexpr int C8 = BN / 8;
  constexpr int RPP = NT / C8;       // rows per pass
  const int c8 = t % C8;
  const int gcol = n0 + c8 * 8;
  if (gcol >= a.cout) return;
  const int epi = a.epilogue;
  const vfml_gfptr addend = addend_of(a);
  f32x4 bias4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (a.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (gcol + e < a.cout) bias4[e >> 2][e & 3] = a.bias[gcol + e];
  }
  // reads 4 channels at (row, col) of an aux operand in either format
  auto aux4 = [&](const float* base, int ld, int64_t row, int col) -> f32x4 {
    f32x4 x;
    if (a.aux16) {
      const char* u = reinterpret_cast<const char*>(base + row * ld + (col & ~7)) + (col & 4) * 2;
      const h16x2 h0 = *reinterpret_cast<const h16x2*>(u), h1 = *reinterpret_cast<const h16x2*>(u + 4);
      const h16x2 l0 = *reinterpret_cast<const h16x2*>(u + 16), l1 = *reinterpret_cast<const h16x2*>(u + 20);
      x[0] = (float)h0[0] + (float)l0[0];
      x[1] = (float)h0[1] + (float)l0[1];
      x[2] = (float)h1[0] + (float)l1[0];
      x[3] = (float)h1[1] + (float)l1[1];
    } else if (a.vec_ok) {
      x = *reinterpret_cast<const f32x4*>(base + row * ld + col);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = base[row * ld + col + e];
    }
    return x;
  };
  for (int row = t / C8; row < nrows; row += RPP) {
    const int grow = m0 + (row >> 5) * rstride + roff + (row & 31);
    if (grow >= a.M) continue;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int col = gcol + 4 * q;
      if (col >= a.cout) break;
      const bool lowhalf = col < a.split;   // split is a multiple of 4: a quad never straddles it
      const int nvalid = a.cout - col >= 4 ? 4 : a.cout - col;
      f32x4 v = *reinterpret_cast<const f32x4*>(&sC[row * LDC + c8 * 8 + 4 * q]);
      f32x4 add4 = {0.f, 0.f, 0.f, 0.f};
      if (a.addend) {
        if (nvalid == 4 && a.vec_ok) {
          add4 = *(vfml_gf4ptr)(addend + (int64_t)grow * a.ld_addend + col);
        } else {
          for (int e = 0; e < nvalid; ++e) add4[e] = addend[(int64_t)grow * a.ld_addend + col + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] * a.w_inv + add4[e] + bias4[q][e]) * a.out_scale;
      f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
      if (nvalid == 4) {
        if (epi == VFML_EPI_GRU_ZR && !lowhalf) x0 = aux4(a.aux0, a.ld_aux0, grow, col - a.split);
        if (epi == VFML_EPI_GRU_Q) {
          x0 = aux4(a.aux0, a.ld_aux0, grow, col);
          x1 = aux4(a.aux1, a.ld_aux1, grow, col);
        }
        if (epi == VFML_EPI_ADD_AUX) x0 = aux4(a.aux0, a.ld_aux0, grow, col);
      } else {
        for (int e = 0; e < nvalid; ++e) {   // ragged tail: f32 operands only (host check)
          if (epi == VFML_EPI_GRU_ZR && !lowhalf) x0[e] = a.aux0[(int64_t)grow * a.ld_aux0 + col + e - a.split];
          if (epi == VFML_EPI_GRU_Q) {
            x0[e] = a.aux0[(int64_t)grow * a.ld_aux0 + col + e];
            x1[e] = a.aux1[(int64_t)grow * a.ld_aux1 + col + e];
          }
          if (epi == VFML_EPI_ADD_AUX) x0[e] = a.aux0[(int64_t)grow * a.ld_aux0 + col + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = epi1(v[e], epi, lowhalf, x0[e], x1[e]);
      if (a.out16) {
        // hi quad at unit + 8q bytes, lo quad at unit + 16 + 8q (cout % 4 == 0, host check)
        U8 hi, lo;
        split4(v, hi, lo, 0);
        char* u = reinterpret_cast<char*>(a.out + (int64_t)grow * a.ldo + gcol) + 8 * q;
        *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hi.v, hi.v, 0, 1, 2, 3));
        *reinterpret_cast<uint2*>(u + 16) = __builtin_bit_cast(uint2, __builtin_shufflevector(lo.v, lo.v, 0, 1, 2, 3));
      } else {
        float* o = a.out + (int64_t)grow * a.ldo + col;
        if (nvalid == 4 && a.vec_ok) {
          *reinterpret_cast<f32x4*>(o) = v;
        } else {
          for (int e = 0; e < nvalid; ++e) o[e] = v[e];
        }
      }
    }
  }
}

// The same rows as epilogue_rows for the calls the update block makes all day: whole 8-channel units (the tile's columns
// lie inside cout, 16-byte aligned operands), split-row aux operands, the epilogue kind a compile-time constant.  One
// thread per (row, unit): two 16-byte LDS reads, 16-byte loads of the addend / aux units, the same expressions as
// epilogue_rows (bit-identical results), two 16-byte stores.  ~3x fewer vector instructions than the general routine - an
// epilogue shares its SIMD with the other resident workgroup's MFMA stream and takes ~10 cycles per vector instruction
// there (profiles/r02_kernel_anatomy.md).  Returns false (nothing done) when the call is not of that shape.
template <int EPI>
__device__ __forceinline__ void epi_unit(f32x4 (&v)[2], const f32x4 (&x0)[2], const f32x4 (&x1)[2], bool lowhalf) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float r = v[q][e];
      if constexpr (EPI == VFML_EPI_RELU) r = fmaxf(r, 0.f);
      else if constexpr (EPI == VFML_EPI_TANH) r = tanhf(r);
      else if constexpr (EPI == VFML_EPI_SIGMOID) r = sigmoidf_(r);
      else if constexpr (EPI == VFML_EPI_GRU_ZR) { r = sigmoidf_(r); r = lowhalf ? r : r * x0[q][e]; }
      else if constexpr (EPI == VFML_EPI_GRU_Q) r = (1.f - x0[q][e]) * x1[q][e] + x0[q][e] * tanhf(r);
      v[q][e] = r;
    }
}

// the 8 channels of a split-row unit at float offset `off` from base (16 B of hi halves, 16 B of lo halves)
__device__ __forceinline__ void load_unit16(const float* base, int64_t off, f32x4 (&x)[2]) {
  const h16x8 h = *reinterpret_cast<const h16x8*>(base + off);
  const h16x8 l = *reinterpret_cast<const h16x8*>(base + off + 4);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e >> 2][e & 3] = (float)h[e] + (float)l[e];
}

template <int BN, int NT, int EPI>
__device__ __forceinline__ void epilogue_rows_fast_k(const SplitArgs& a, const float* sC, int m0, int n0, int t, int nrows,
                                                     int rstride, int roff) {
  constexpr int LDC = BN + 4;
  constexpr int C8 = BN / 8;
  constexpr int RPP = NT / C8;
  const int c8 = t % C8;
  const int gcol = n0 + c8 * 8;
  if (gcol >= a.cout) return;
  const bool whole = gcol + 8 <= a.cout;        // cout % 4 == 0: the last unit may be its first quad only
  f32x4 bias4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (a.bias) {
    bias4[0] = *reinterpret_cast<const f32x4*>(a.bias + gcol);
    if (whole) bias4[1] = *reinterpret_cast<const f32x4*>(a.bias + gcol + 4);
  }
  const bool lowhalf = gcol < a.split;          // split is a multiple of 8 here: a unit never straddles it
  const vfml_gfptr addend = addend_of(a);
  // The addend / aux units of a row are requested BEFORE any store of its batch: vmcnt counts in order, a load issued behind
  // a store could only be waited for together with the store's acknowledgement.
  struct RowOps {
    f32x4 add[2], x0[2], x1[2];
  };
  auto load_row = [&](int row, RowOps& o) {
    const int grow = m0 + (row >> 5) * rstride + roff + (row & 31);
    o.add[0] = o.add[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    o.x0[0] = o.x0[1] = o.x1[0] = o.x1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row >= nrows || grow >= a.M) return;
    if (a.addend) {
      o.add[0] = *(vfml_gf4ptr)(addend + (int64_t)grow * a.ld_addend + gcol);
      if (whole) o.add[1] = *(vfml_gf4ptr)(addend + (int64_t)grow * a.ld_addend + gcol + 4);
    }
    if constexpr (EPI == VFML_EPI_GRU_ZR) {
      if (!lowhalf) load_unit16(a.aux0, (int64_t)grow * a.ld_aux0 + gcol - a.split, o.x0);
    }
    if constexpr (EPI == VFML_EPI_GRU_Q) {
      load_unit16(a.aux0, (int64_t)grow * a.ld_aux0 + gcol, o.x0);
      load_unit16(a.aux1, (int64_t)grow * a.ld_aux1 + gcol, o.x1);
    }
  };
  // EB rows per thread in flight: the addend / aux loads of a batch all go out before its first row is touched.  (Round 3:
  // with ONE row requested ahead, every row of a GRU epilogue waited out a whole memory round trip - the work between a
  // load and its use was ~200 cycles - and the four gate convolutions of an iteration ran 24-40 us longer than the same
  // shapes with a plain ReLU epilogue: tools/exp/r03_l2_touch/README.md.  A 64-row slab is one batch per thread.)
  // (the q gate's rows carry three operands, 24 registers each, beside the accumulators of the slabs still to come: two)
  constexpr int EB = EPI == VFML_EPI_GRU_Q ? 2 : 4;
  for (int row0 = t / C8; row0 < nrows; row0 += EB * RPP) {
    RowOps ops[EB];
    static_for<EB>([&](auto bc) { load_row(row0 + decltype(bc)::value * RPP, ops[decltype(bc)::value]); });
    static_for<EB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      const int row = row0 + b * RPP;
      const int grow = m0 + (row >> 5) * rstride + roff + (row & 31);
      if (row < nrows && grow < a.M) {
        const RowOps& cur = ops[b];
        f32x4 v[2];
        v[0] = *reinterpret_cast<const f32x4*>(&sC[row * LDC + c8 * 8]);
        v[1] = *reinterpret_cast<const f32x4*>(&sC[row * LDC + c8 * 8 + 4]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] = (v[q][e] * a.w_inv + cur.add[q][e] + bias4[q][e]) * a.out_scale;
        epi_unit<EPI>(v, cur.x0, cur.x1, lowhalf);
        if (a.out16) {
          U8 hi, lo;
          split4(v[0], hi, lo, 0);
          split4(v[1], hi, lo, 4);
          float* u = a.out + (int64_t)grow * a.ldo + gcol;
          if (whole) {
            *reinterpret_cast<h16x8*>(u) = hi.v;
            *reinterpret_cast<h16x8*>(u + 4) = lo.v;
          } else {       // the unit's second quad belongs to someone else (the motion features' flow channels)
            *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hi.v, hi.v, 0, 1, 2, 3));
            *reinterpret_cast<uint2*>(u + 4) = __builtin_bit_cast(uint2, __builtin_shufflevector(lo.v, lo.v, 0, 1, 2, 3));
          }
        } else {
          float* o = a.out + (int64_t)grow * a.ldo + gcol;
          *reinterpret_cast<f32x4*>(o) = v[0];
          if (whole) *reinterpret_cast<f32x4*>(o + 4) = v[1];
        }
      }
    });
  }
}

// workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every outstanding global store
// (vmcnt(0)), which serialises an epilogue's slabs on the store round trip
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Projection epilogue (include/vfml.h vfml_conv_desc.proj_out; the flow head: 3x3 to 256 channels, ReLU, then 256 -> 4 over
// 3x3 run as a 1x1 to 36 tap-major columns).  A 64-row slab of the workgroup's fp32 tile is in LDS (sC, row stride BN + 4
// floats, BN = 128): every (row, 8-channel unit) becomes relu((acc * w_inv + bias) * out_scale) as 16 bytes of hi halves
// + 16 bytes of lo halves IN PLACE (the unit's own 32 bytes - exactly what a split-row store would have written to HBM);
// then wave w multiplies rows 16 w .. 16 w + 15 by the tile's 128-channel slice of the projection weights on the matrix
// cores (the full split product, 16x16x32 MFMAs, weight fragments from the copy proj_weights_store left behind the slab) and
// stores the [16][proj_n] partial sums of this column tile.  The 256-channel map never travels to HBM and back.
// The tile's slice of the projection weights - 48 rows (those past proj_n zero) x 128 channels x {hi, lo} = 24 KB - goes to
// LDS once per tile, behind the slab (rows of 256 + 16 bytes: conflict-free fragment reads): requested into registers
// before the slab loop (proj_weights_load), written once the K loop's stages are free (proj_weights_store).
constexpr int PROJ_ROW = 272, PROJ_ROWS = 48, PROJ_PIECES = 2 * PROJ_ROWS * 16 / 256;
template <int BN>
constexpr int proj_lds_off() { return 64 * (BN + 4) * 4; }
__device__ __forceinline__ void proj_weights_load(const SplitArgs& a, int n0, int t, u32x4 (&pw)[PROJ_PIECES]) {
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.proj_w), 0, a.proj_bytes, 0x00020000);
  static_for<PROJ_PIECES>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int q = t + 256 * k, plane = q / (PROJ_ROWS * 16), rem = q % (PROJ_ROWS * 16), n = rem >> 4, seg = rem & 15;
    const int off = n < a.proj_n ? (n * a.proj_kp + n0) * 2 + seg * 16 : (int)0x40000000;
    pw[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, plane ? a.proj_lo_off : 0, 0));
  });
}
__device__ __forceinline__ void proj_weights_store(const u32x4 (&pw)[PROJ_PIECES], char* lds, int t) {
  static_for<PROJ_PIECES>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int q = t + 256 * k, plane = q / (PROJ_ROWS * 16), rem = q % (PROJ_ROWS * 16), n = rem >> 4, seg = rem & 15;
    *reinterpret_cast<u32x4*>(lds + (plane * PROJ_ROWS + n) * PROJ_ROW + seg * 16) = pw[k];
  });
}

// AHI (the "2a" product: activations as plain f16): the lo halves of relu(out) take no part - d = bh*ah + bl*ah.
template <int BN, int NT, bool AHI>
__device__ __forceinline__ void epilogue_proj_slab(const SplitArgs& a, float* sC, int m0, int n0, int t, int nrows, int rstride,
                                                   int roff) {
  static_assert(BN == 128 && NT == 256, "projection epilogue: 128-column tiles of four waves");
  constexpr int LDC = BN + 4, C8 = BN / 8, RPP = NT / C8;
  const int c8 = t % C8;
  const int gcol = n0 + c8 * 8;
  f32x4 bias4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (a.bias && gcol < a.cout) {
    bias4[0] = *reinterpret_cast<const f32x4*>(a.bias + gcol);
    bias4[1] = *reinterpret_cast<const f32x4*>(a.bias + gcol + 4);
  }
  for (int row = t / C8; row < nrows; row += RPP) {
    float* u = &sC[row * LDC + c8 * 8];
    f32x4 v[2] = {*reinterpret_cast<const f32x4*>(u), *reinterpret_cast<const f32x4*>(u + 4)};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q][e] = fmaxf((v[q][e] * a.w_inv + bias4[q][e]) * a.out_scale, 0.f);
    if (gcol >= a.cout) v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    U8 hi, lo;
    split4(v[0], hi, lo, 0);
    split4(v[1], hi, lo, 4);
    *reinterpret_cast<h16x8*>(u) = hi.v;
    *reinterpret_cast<h16x8*>(u + 4) = lo.v;
  }
  lds_barrier();
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 15, g = lane >> 4;
  f32x4 d[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const char* arow = reinterpret_cast<const char*>(sC) + (16 * wave + r) * (LDC * 4) + g * 32;
  const char* brow = reinterpret_cast<const char*>(sC) + proj_lds_off<BN>() + r * PROJ_ROW + g * 16;
  static_for<4>([&](auto sc) {
    constexpr int ks = decltype(sc)::value;            // 32 channels of the tile per step
    h16x8 bh[3], bl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bh[j] = *reinterpret_cast<const h16x8*>(brow + 16 * j * PROJ_ROW + ks * 64);
      bl[j] = *reinterpret_cast<const h16x8*>(brow + (PROJ_ROWS + 16 * j) * PROJ_ROW + ks * 64);
    }
    const h16x8 ah = *reinterpret_cast<const h16x8*>(arow + ks * 128);
    h16x8 al;
    if constexpr (!AHI) al = *reinterpret_cast<const h16x8*>(arow + ks * 128 + 16);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah, d[j], 0, 0, 0);
      d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah, d[j], 0, 0, 0);
      if constexpr (!AHI) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al, d[j], 0, 0, 0);
    }
  });
  // lane: pixel row r of the wave's 16, projection columns 16 j + 4 g .. + 3
  const int srow = 16 * wave + r;
  const int grow = m0 + (srow >> 5) * rstride + roff + (srow & 31);
  if (srow < nrows && grow < a.M) {
    float* o = a.proj_out + ((int64_t)(n0 / BN) * a.M + grow) * a.ld_proj;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int n = 16 * j + 4 * g;
      if (n < a.proj_n) {
        f32x4 v = d[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a.proj_inv;
        *reinterpret_cast<f32x4*>(o + n) = v;
      }
    }
  }
}

template <int BN, int NT>
__device__ __forceinline__ bool epilogue_rows_fast(const SplitArgs& a, const float* sC, int m0, int n0, int t, int nrows,
                                                   int rstride, int roff) {
  // (uniform over the workgroup: every thread takes the same route)
  const bool gru = a.epilogue == VFML_EPI_GRU_ZR || a.epilogue == VFML_EPI_GRU_Q;
  if (!(a.fast_epi && a.vec_ok && a.cout % 4 == 0 && (!gru || (a.aux16 && a.split % 8 == 0 && a.cout % 8 == 0)) &&
        (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15u) == 0) && a.epilogue != VFML_EPI_TANH_RELU &&
        a.epilogue != VFML_EPI_ADD_AUX))
    return false;
  switch (a.epilogue) {
    case VFML_EPI_NONE: epilogue_rows_fast_k<BN, NT, VFML_EPI_NONE>(a, sC, m0, n0, t, nrows, rstride, roff); return true;
    case VFML_EPI_RELU: epilogue_rows_fast_k<BN, NT, VFML_EPI_RELU>(a, sC, m0, n0, t, nrows, rstride, roff); return true;
    case VFML_EPI_GRU_ZR: epilogue_rows_fast_k<BN, NT, VFML_EPI_GRU_ZR>(a, sC, m0, n0, t, nrows, rstride, roff); return true;
    case VFML_EPI_GRU_Q: epilogue_rows_fast_k<BN, NT, VFML_EPI_GRU_Q>(a, sC, m0, n0, t, nrows, rstride, roff); return true;
    default: return false;
  }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes, char* lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  // address = base + voffset + soffset; only voffset is range-checked (an out-of-range lane writes zeros)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voff_bytes, soff_bytes, 0, 0);
#endif
}

// bank swizzle of the LDS image read by 16x16x32 fragments (conv_gemm_dma_kernel, MF16)
__device__ __forceinline__ constexpr int swz16(int x) { return x ^ ((((x >> 1) ^ (x >> 2)) & 1) << 1); }

}  // namespace
