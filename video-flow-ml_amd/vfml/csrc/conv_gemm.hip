// Implicit-GEMM convolution / NT-GEMM on the CDNA4 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[p][co] = epi( sum_k A[p][k] * W[co][k] ),   k = (ky*KW + kx)*Ctot + ci
//
// A is never materialised: a row of A is the (ky,kx)-shifted NHWC pixel of up to two
// channel-concatenated sources, gathered 16 B (4 channels) at a time, zero outside the image.
// Exact f32: the MFMA is a k-ordered fmaf chain, so results track a CPU fp32 conv to rounding.
//
// Tiling: 128 pixels x BN output channels per 256-thread workgroup (4 wave64), K stepped by 32.
// LDS image per operand: [k/4][row] float4 (row stride padded by one float4), so that
//   - the staging write (8 lanes = 8 k-groups of one row) lands on 8 distinct 4-bank slots,
//   - a wave's fragment read is 32 consecutive float4 (conflict-free ds_read_b128), and one
//     read feeds 4 MFMAs: lane half h takes k-group 2*kq+h, element j feeds MFMA j.  The
//     k-permutation is the same for A and B, so the sum over k is unchanged.
// Two LDS buffers, one barrier per K step; next tile's global loads are issued before the
// MFMAs of the current one.
#include "vfml_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int KG = BK / 4;  // float4 k-groups per K step

struct ConvArgs {
  const float* in0; const float* in1;
  const float* weight; const float* bias;
  const float* aux0; const float* aux1; const float* addend;
  float* out;
  int c0, ld0, c1, ld1, ctot, ld_addend;
  int H, W, ho, wo;
  int kw, stride, pad_h, pad_w;
  int M, K, cout;
  int ldo, ld_aux0, ld_aux1;
  int epilogue, split;
  float out_scale;
  int mtiles, ntiles;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvArgs a) {
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AROWS = BM / 32;  // float4 loads per thread per K step (A)
  constexpr int BROWS = BN / 32;
  constexpr int SA = KG * (BM + 1);  // float4 per A buffer
  constexpr int SB = KG * (BN + 1);

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* sA = reinterpret_cast<f32x4*>(smem_raw);
  f32x4* sB = sA + 2 * SA;

  // XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so neighbouring pixel tiles / the same weight panel stay in one L2.
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
  const int kg = t & 7;
  const int lrow = t >> 3;  // 0..31

  // Per-thread gather bases for its A rows (fixed for the whole K loop).
  int iy0[AROWS], ix0[AROWS];
  int64_t pbase[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < a.M) {
      const int hw = a.ho * a.wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / a.wo;
      const int ox = rem - oy * a.wo;
      iy0[i] = oy * a.stride - a.pad_h;
      ix0[i] = ox * a.stride - a.pad_w;
      pbase[i] = (int64_t)n * a.H * a.W;
    } else {
      iy0[i] = -(1 << 28);
      ix0[i] = 0;
      pbase[i] = 0;
    }
  }

  f32x4 ra[AROWS], rb[BROWS];

  auto load_tile = [&](int k0) {
    const int k = k0 + kg * 4;
    const bool kok = k < a.K;
    int tap = 0, c = 0;
    if (kok) {
      tap = k / a.ctot;
      c = k - tap * a.ctot;
    }
    const int ky = tap / a.kw;
    const int kx = tap - ky * a.kw;
    const float* src = a.in0;
    int ld = a.ld0;
    if (c >= a.c0) {
      src = a.in1;
      ld = a.ld1;
      c -= a.c0;
    }
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int iy = iy0[i] + ky;
      const int ix = ix0[i] + kx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        const int64_t off = (pbase[i] + (int64_t)iy * a.W + ix) * ld + c;
        v = *reinterpret_cast<const f32x4*>(src + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int col = n0 + lrow + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && col < a.cout) v = *reinterpret_cast<const f32x4*>(a.weight + (int64_t)col * a.K + k);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) sA[buf * SA + kg * (BM + 1) + lrow + 32 * i] = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) sB[buf * SB + kg * (BN + 1) + lrow + 32 * i] = rb[i];
  };

  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int r = lane & 31;
  const int half = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    const f32x4* pa = sA + buf * SA + wm * (BM / WM) + r;
    const f32x4* pb = sB + buf * SB + wn * (BN / WN) + r;
#pragma unroll
    for (int kq = 0; kq < KG / 2; ++kq) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = pa[(2 * kq + half) * (BM + 1) + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = pb[(2 * kq + half) * (BN + 1) + j * 32];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // Epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
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
        float v = acc[i][j][e];
        if (a.addend) v += a.addend[(int64_t)row * a.ld_addend + col];
        v = (v + bias) * a.out_scale;
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
          case VFML_EPI_ADD_AUX: v += a.aux0[(int64_t)row * a.ld_aux0 + col]; break;
          default: break;
        }
        a.out[(int64_t)row * a.ldo + col] = v;
      }
    }
  }
}

template <int BN, int WM, int WN>
int launch(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (KG * (BM + 1) + KG * (BN + 1)) * sizeof(f32x4);
  static bool attr_done = false;  // per instantiation; raising the dynamic-LDS cap is idempotent
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BN, WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vfml_set_error("vfml_conv2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_kernel<BN, WM, WN>), dim3(a.mtiles * a.ntiles), dim3(256), lds, s, a);
  return vfml_check_launch("vfml_conv2d");
}

}  // namespace

extern "C" int vfml_conv2d(const vfml_conv_desc* d, void* stream) {
  VFML_REQUIRE(d != nullptr, "vfml_conv2d: null descriptor");
  VFML_REQUIRE(d->out_t == nullptr && d->flags == 0, "vfml_conv2d: out_t / flags are vfml_conv2d_split (GEMM form) features");
  VFML_REQUIRE(d->stats_part == nullptr, "vfml_conv2d: stats_part is a vfml_conv2d_split feature");
  VFML_REQUIRE(d->in0 && d->weight && d->out, "vfml_conv2d: null in0/weight/out");
  VFML_REQUIRE(d->c0 > 0 && d->c0 % 4 == 0 && d->ld0 % 4 == 0 && d->ld0 >= d->c0,
               "vfml_conv2d: c0=%d ld0=%d must be multiples of 4 with ld0>=c0", d->c0, d->ld0);
  const bool two = d->in1 != nullptr;
  VFML_REQUIRE(two ? (d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && d->ld1 >= d->c1) : d->c1 == 0,
               "vfml_conv2d: c1=%d ld1=%d inconsistent with in1", d->c1, d->ld1);
  VFML_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "vfml_conv2d: empty problem");
  VFML_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "vfml_conv2d: bad kernel geometry");
  VFML_REQUIRE(d->ldo >= d->cout, "vfml_conv2d: ldo=%d < cout=%d", d->ldo, d->cout);
  VFML_REQUIRE(vfml_aligned16(d->in0) && vfml_aligned16(d->weight) && (!two || vfml_aligned16(d->in1)),
               "vfml_conv2d: in0/in1/weight must be 16-byte aligned");
  const int ho = (d->h + 2 * d->pad_h - d->kh) / d->stride + 1;
  const int wo = (d->w + 2 * d->pad_w - d->kw) / d->stride + 1;
  VFML_REQUIRE(ho > 0 && wo > 0, "vfml_conv2d: empty output");
  const int64_t M64 = (int64_t)d->n * ho * wo;
  VFML_REQUIRE(M64 < (1ll << 31) - BM, "vfml_conv2d: too many output pixels");
  if (d->epilogue == VFML_EPI_GRU_ZR)
    VFML_REQUIRE(d->aux0 && d->split > 0 && d->split < d->cout, "vfml_conv2d: GRU_ZR needs aux0 and split");
  if (d->epilogue == VFML_EPI_GRU_Q) VFML_REQUIRE(d->aux0 && d->aux1, "vfml_conv2d: GRU_Q needs aux0 and aux1");
  VFML_REQUIRE(d->epilogue >= VFML_EPI_NONE && d->epilogue <= VFML_EPI_ADD_AUX, "vfml_conv2d: bad epilogue");

  ConvArgs a;
  a.in0 = d->in0; a.in1 = two ? d->in1 : d->in0;
  a.weight = d->weight; a.bias = d->bias; a.aux0 = d->aux0; a.aux1 = d->aux1; a.out = d->out;
  a.addend = d->addend; a.ld_addend = d->ld_addend;
  VFML_REQUIRE(!d->addend_ind, "vfml_conv2d: addend_ind is implemented by vfml_conv2d_split only");
  VFML_REQUIRE(!d->addend || d->ld_addend >= d->cout, "vfml_conv2d: ld_addend=%d < cout", d->ld_addend);
  a.c0 = d->c0; a.ld0 = d->ld0; a.c1 = d->c1; a.ld1 = two ? d->ld1 : d->ld0; a.ctot = d->c0 + d->c1;
  a.H = d->h; a.W = d->w; a.ho = ho; a.wo = wo;
  a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.M = (int)M64; a.K = d->kh * d->kw * a.ctot; a.cout = d->cout;
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
