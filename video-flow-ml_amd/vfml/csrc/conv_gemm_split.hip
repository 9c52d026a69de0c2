// Implicit-GEMM convolution / NT-GEMM on the CDNA4 f16 matrix cores with fp32-grade accuracy
// ("split-f16": every operand x is carried as hi = f16(x) and lo = f16((x - hi) * 2^11), and
//   a*b ~= a_hi*b_hi + 2^-11 * (a_hi*b_lo + a_lo*b_hi)      (a_lo*b_lo ~ 2^-22 |ab| is dropped)
// f16 x f16 products are exact in the f32 accumulator, so the result carries ~22 mantissa bits:
// within a few ulp of an f32 conv, at 3 v_mfma_f32_32x32x16_f16 (1024 FLOP/clk/SIMD each) per
// product instead of one v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD): 16/3 = 5.3x the f32 MFMA rate.
// Scaling lo by 2^11 keeps it a normal f16 whenever hi is, so nothing depends on denormals.
//
// Activations stay fp32 NHWC in HBM and are split while being staged into LDS; weights (and the
// pooled target features of the correlation GEMM) are pre-split once by vfml_split_f16 into two
// f16 planes [cout][Kp], Kp = K rounded up to 8, zero padded.
//
// Tiling: 128 pixels x BN channels per 256-thread workgroup, K stepped by 32.  LDS image per
// operand plane: [k/8][row] 16-byte units (8 halves), row stride padded by 2 units: the staging
// write of 8 consecutive lanes (4 k-groups x 2 rows) covers all 32 banks, a wave's fragment read is
// 32 consecutive units (conflict-free ds_read_b128) and is exactly one MFMA operand
// (lane l: row l&31, k = 8*(l>>5)+j).
#include <hip/hip_fp16.h>
#include "vfml_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int KG = BK / 8;  // 16-byte units (8 halves) per row per K step
constexpr float LO_SCALE = 2048.0f;
constexpr float LO_INV = 1.0f / 2048.0f;

struct SplitArgs {
  const float* in0; const float* in1;
  const _Float16* whi; const _Float16* wlo; const float* bias;
  const float* aux0; const float* aux1;
  float* out;
  int c0, ld0, c1, ld1, ctot;
  int H, W, ho, wo;
  int kw, stride, pad_h, pad_w;
  int M, K, Kp, cout;
  int ldo, ld_aux0, ld_aux1;
  int epilogue, split;
  float out_scale;
  int mtiles, ntiles;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

union U8 {
  h16x8 v;
  h16x2 p[4];
};

// x (4 floats) -> hi/lo halves written at element offset `at` (0 or 4) of the 8-wide units
__device__ __forceinline__ void split4(const f32x4 x, U8& hi, U8& lo, int at) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float a = x[2 * e], b = x[2 * e + 1];
    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ra = (a - (float)h[0]) * LO_SCALE;
    const float rb = (b - (float)h[1]) * LO_SCALE;
    const fp16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    hi.p[at / 2 + e] = __builtin_bit_cast(h16x2, h);
    lo.p[at / 2 + e] = __builtin_bit_cast(h16x2, l);
  }
}

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_gemm_split_kernel(const SplitArgs a) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AU = BM / 64;   // A units per thread per K step (each unit: 8 k of one row)
  constexpr int BU = BN / 64;   // B units per thread per K step; BN=32 -> threads >= 128 idle on B
  constexpr int BUN = BU > 0 ? BU : 1;
  constexpr int RSA = BM + 2;   // unit row stride per k-group (padded)
  constexpr int RSB = BN + 2;
  constexpr int SA = KG * RSA;  // units per A plane per buffer
  constexpr int SB = KG * RSB;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  h16x8* sAh = reinterpret_cast<h16x8*>(smem_raw);
  h16x8* sAl = sAh + 2 * SA;
  h16x8* sBh = sAl + 2 * SA;
  h16x8* sBl = sBh + 2 * SB;

  const int nwg = a.mtiles * a.ntiles;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = bid % a.ntiles;
  const int mt = bid / a.ntiles;
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  const int t = threadIdx.x;
  const int kg = t & 3;
  const int lrow = t >> 2;  // 0..63

  // Per-row gather state, fixed for the whole K loop.  All A offsets are 32-bit element offsets
  // (the host checks that every source spans < 2^31 floats).
  int iy0[AU], ix0[AU], rp0[AU], rp1[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int m = m0 + lrow + 64 * i;
    if (m < a.M) {
      const int hw = a.ho * a.wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / a.wo;
      const int ox = rem - oy * a.wo;
      iy0[i] = oy * a.stride - a.pad_h;
      ix0[i] = ox * a.stride - a.pad_w;
      const int pix = (n * a.H + iy0[i]) * a.W + ix0[i];
      rp0[i] = pix * a.ld0;
      rp1[i] = pix * a.ld1;
    } else {
      iy0[i] = -(1 << 28);
      ix0[i] = 0;
      rp0[i] = rp1[i] = 0;
    }
  }
  // Per-half k state (k = k0 + kg*8 + hf*4): channel c within the tap, tap position (ky, kx);
  // advanced incrementally by BK per K step (no divisions in the loop when ctot >= BK).
  int kc[2], kky[2], kkx[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int k = kg * 8 + hf * 4;
    const int tap = k / a.ctot;
    kc[hf] = k - tap * a.ctot;
    kky[hf] = tap / a.kw;
    kkx[hf] = tap - kky[hf] * a.kw;
  }
  const bool big_c = a.ctot >= BK;
  int colKp[BUN];
#pragma unroll
  for (int i = 0; i < BUN; ++i) colKp[i] = (n0 + lrow + 64 * i) * a.Kp;   // cout * Kp < 2^31 (host check)

  f32x4 ra[AU][2];
  h16x8 rbh[BUN], rbl[BUN];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const bool kok = k0 + kg * 8 + hf * 4 < a.K;
      int c = kc[hf];
      const int ky = kky[hf], kx = kkx[hf];
      const bool s1 = c >= a.c0;
      const float* src = s1 ? a.in1 : a.in0;
      const int ld = s1 ? a.ld1 : a.ld0;
      if (s1) c -= a.c0;
      const int tapoff = (ky * a.W + kx) * ld + c;
#pragma unroll
      for (int i = 0; i < AU; ++i) {
        const int iy = iy0[i] + ky;
        const int ix = ix0[i] + kx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *reinterpret_cast<const f32x4*>(src + ((s1 ? rp1[i] : rp0[i]) + tapoff));
        ra[i][hf] = v;
      }
      // advance to the next K step
      if (big_c) {
        int cn = kc[hf] + BK;
        if (cn >= a.ctot) {
          cn -= a.ctot;
          if (++kkx[hf] == a.kw) {
            kkx[hf] = 0;
            ++kky[hf];
          }
        }
        kc[hf] = cn;
      } else {
        const int k = k0 + BK + kg * 8 + hf * 4;
        const int tap = k / a.ctot;
        kc[hf] = k - tap * a.ctot;
        kky[hf] = tap / a.kw;
        kkx[hf] = tap - kky[hf] * a.kw;
      }
    }
    const int kb = k0 + kg * 8;
#pragma unroll
    for (int i = 0; i < BUN; ++i) {
      const int col = n0 + lrow + 64 * i;
      h16x8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
      if ((BU > 0 || lrow < BN) && kb < a.Kp && col < a.cout) {
        vh = *reinterpret_cast<const h16x8*>(a.whi + (colKp[i] + kb));
        vl = *reinterpret_cast<const h16x8*>(a.wlo + (colKp[i] + kb));
      }
      rbh[i] = vh;
      rbl[i] = vl;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      U8 hi, lo;
      split4(ra[i][0], hi, lo, 0);
      split4(ra[i][1], hi, lo, 4);
      sAh[buf * SA + kg * RSA + lrow + 64 * i] = hi.v;
      sAl[buf * SA + kg * RSA + lrow + 64 * i] = lo.v;
    }
#pragma unroll
    for (int i = 0; i < BUN; ++i) {
      if (BU > 0 || lrow < BN) {
        sBh[buf * SB + kg * RSB + lrow + 64 * i] = rbh[i];
        sBl[buf * SB + kg * RSB + lrow + 64 * i] = rbl[i];
      }
    }
  };

  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int r = lane & 31;
  const int half = lane >> 5;

  f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        accx[i][j][e] = 0.f;
      }

  const int nk = (a.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    const int oa = buf * SA + wm * (BM / WM) + r;
    const int ob = buf * SB + wn * (BN / WN) + r;
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = sAh[oa + (2 * ks + half) * RSA + i * 32];
        al[i] = sAl[oa + (2 * ks + half) * RSA + i * 32];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = sBh[ob + (2 * ks + half) * RSB + j * 32];
        bl[j] = sBl[ob + (2 * ks + half) * RSB + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / WN) + j * 32 + r;
    if (col >= a.cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row >= a.M) continue;
        float v = (acc[i][j][e] + accx[i][j][e] * LO_INV + bias) * a.out_scale;
        switch (a.epilogue) {
          case VFML_EPI_RELU: v = fmaxf(v, 0.f); break;
          case VFML_EPI_TANH: v = tanhf(v); break;
          case VFML_EPI_SIGMOID: v = sigmoidf_(v); break;
          case VFML_EPI_TANH_RELU: v = col < a.split ? tanhf(v) : fmaxf(v, 0.f); break;
          case VFML_EPI_GRU_ZR:
            v = sigmoidf_(v);
            if (col >= a.split) v *= a.aux0[(int64_t)row * a.ld_aux0 + (col - a.split)];
            break;
          case VFML_EPI_GRU_Q: {
            const float z = a.aux0[(int64_t)row * a.ld_aux0 + col];
            const float h = a.aux1[(int64_t)row * a.ld_aux1 + col];
            v = (1.f - z) * h + z * tanhf(v);
          } break;
          default: break;
        }
        a.out[(int64_t)row * a.ldo + col] = v;
      }
    }
  }
}

template <int BN, int WM, int WN>
int launch(const SplitArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * 2 * (KG * (BM + 2) + KG * (BN + 2)) * 16;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_split_kernel<BN, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vfml_set_error("vfml_conv2d_split: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_split_kernel<BN, WM, WN>), dim3(a.mtiles * a.ntiles), dim3(256), lds, s, a);
  return vfml_check_launch("vfml_conv2d_split");
}

// f32 [rows][k] (row stride ld) -> hi/lo f16 planes [rows][kp], zero padded to kp
__global__ void split_f16_kernel(const float* __restrict__ src, int64_t rows, int k, int ld, int kp,
                                 _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int64_t total = rows * (kp / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rrow = i / (kp / 2);
    const int c = (int)(i - rrow * (kp / 2)) * 2;
    const float a = c < k ? src[rrow * ld + c] : 0.f;
    const float b = c + 1 < k ? src[rrow * ld + c + 1] : 0.f;
    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const fp16x2 l = __builtin_amdgcn_cvt_pkrtz((a - (float)h[0]) * LO_SCALE, (b - (float)h[1]) * LO_SCALE);
    *reinterpret_cast<fp16x2*>(hi + rrow * kp + c) = h;
    *reinterpret_cast<fp16x2*>(lo + rrow * kp + c) = l;
  }
}

}  // namespace

extern "C" int vfml_split_f16(const float* src, int64_t rows, int k, int ld, void* hi, void* lo, int kp,
                              void* stream) {
  VFML_REQUIRE(src && hi && lo, "vfml_split_f16: null pointer");
  VFML_REQUIRE(rows > 0 && k > 0 && ld >= k && kp >= k && kp % 8 == 0, "vfml_split_f16: bad rows/k/ld/kp (kp%%8==0)");
  VFML_REQUIRE(vfml_aligned16(hi) && vfml_aligned16(lo), "vfml_split_f16: hi/lo must be 16-byte aligned");
  const int64_t total = rows * (kp / 2);
  int64_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(split_f16_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, rows, k,
                     ld, kp, (_Float16*)hi, (_Float16*)lo);
  return vfml_check_launch("vfml_split_f16");
}

extern "C" int vfml_conv2d_split(const vfml_conv_desc* d, const void* w_hi, const void* w_lo, int kp, void* stream) {
  VFML_REQUIRE(d != nullptr, "vfml_conv2d_split: null descriptor");
  VFML_REQUIRE(d->in0 && w_hi && w_lo && d->out, "vfml_conv2d_split: null in0/w_hi/w_lo/out");
  VFML_REQUIRE(d->c0 > 0 && d->c0 % 4 == 0 && d->ld0 % 4 == 0 && d->ld0 >= d->c0,
               "vfml_conv2d_split: c0=%d ld0=%d must be multiples of 4 with ld0>=c0", d->c0, d->ld0);
  const bool two = d->in1 != nullptr;
  VFML_REQUIRE(two ? (d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && d->ld1 >= d->c1) : d->c1 == 0,
               "vfml_conv2d_split: c1=%d ld1=%d inconsistent with in1", d->c1, d->ld1);
  VFML_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "vfml_conv2d_split: empty problem");
  VFML_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "vfml_conv2d_split: bad kernel geometry");
  VFML_REQUIRE(d->ldo >= d->cout, "vfml_conv2d_split: ldo=%d < cout=%d", d->ldo, d->cout);
  VFML_REQUIRE(vfml_aligned16(d->in0) && vfml_aligned16(w_hi) && vfml_aligned16(w_lo) && (!two || vfml_aligned16(d->in1)),
               "vfml_conv2d_split: in0/in1/w_hi/w_lo must be 16-byte aligned");
  const int K = d->kh * d->kw * (d->c0 + d->c1);
  VFML_REQUIRE(kp >= K && kp % 8 == 0 && kp < K + 8, "vfml_conv2d_split: kp=%d must be K=%d rounded up to 8", kp, K);
  const int ho = (d->h + 2 * d->pad_h - d->kh) / d->stride + 1;
  const int wo = (d->w + 2 * d->pad_w - d->kw) / d->stride + 1;
  VFML_REQUIRE(ho > 0 && wo > 0, "vfml_conv2d_split: empty output");
  const int64_t M64 = (int64_t)d->n * ho * wo;
  VFML_REQUIRE(M64 < (1ll << 31) - BM, "vfml_conv2d_split: too many output pixels");
  {
    const int64_t px = (int64_t)d->n * d->h * d->w + (int64_t)(d->pad_h + 1) * d->w + d->pad_w;
    VFML_REQUIRE(px * d->ld0 < (1ll << 31) && (!two || px * d->ld1 < (1ll << 31)),
                 "vfml_conv2d_split: a source spans >= 2^31 floats");
    VFML_REQUIRE(((int64_t)d->cout + 128) * kp < (1ll << 31), "vfml_conv2d_split: weight planes too large");
  }
  if (d->epilogue == VFML_EPI_GRU_ZR)
    VFML_REQUIRE(d->aux0 && d->split > 0 && d->split < d->cout, "vfml_conv2d_split: GRU_ZR needs aux0 and split");
  if (d->epilogue == VFML_EPI_GRU_Q) VFML_REQUIRE(d->aux0 && d->aux1, "vfml_conv2d_split: GRU_Q needs aux0 and aux1");
  VFML_REQUIRE(d->epilogue >= VFML_EPI_NONE && d->epilogue <= VFML_EPI_GRU_Q, "vfml_conv2d_split: bad epilogue");

  SplitArgs a;
  a.in0 = d->in0; a.in1 = two ? d->in1 : d->in0;
  a.whi = (const _Float16*)w_hi; a.wlo = (const _Float16*)w_lo; a.bias = d->bias;
  a.aux0 = d->aux0; a.aux1 = d->aux1; a.out = d->out;
  a.c0 = d->c0; a.ld0 = d->ld0; a.c1 = d->c1; a.ld1 = two ? d->ld1 : d->ld0; a.ctot = d->c0 + d->c1;
  a.H = d->h; a.W = d->w; a.ho = ho; a.wo = wo;
  a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.M = (int)M64; a.K = K; a.Kp = kp; a.cout = d->cout;
  a.ldo = d->ldo; a.ld_aux0 = d->ld_aux0; a.ld_aux1 = d->ld_aux1;
  a.epilogue = d->epilogue; a.split = d->split; a.out_scale = d->out_scale;
  a.mtiles = (a.M + BM - 1) / BM;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d->cout > 64) {
    a.ntiles = (d->cout + 127) / 128;
    return launch<128, 2, 2>(a, s);
  } else if (d->cout > 32) {
    a.ntiles = 1;
    return launch<64, 2, 2>(a, s);
  }
  a.ntiles = 1;
  return launch<32, 4, 1>(a, s);
}
