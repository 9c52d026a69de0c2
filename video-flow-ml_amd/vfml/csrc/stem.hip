// The encoders' stem: 7x7 convolution, stride 2, padding 3, over the 4-channel NHWC4 frame (RGB0) to 64 channels (K2's first
// layer, `fnet.conv1` / `cnet.conv1`), split-f16 arithmetic (three MFMAs per product), plus the instance-norm partial sums of
// its result.  On the general register-staged kernel this layer ran a 32-deep K step over 4 channels x 8 taps with every tap a
// separate gather (177 us per 1080p frame, 96 TFLOP/s algorithmic).  Here a workgroup owns an 8 x 64 tile of OUTPUT pixels and
// holds everything it needs in LDS for the whole tile - no staging loop, no barrier inside the K loop:
//   patch    the (2*8 + 5) x (2*64 + 5) input pixels under the tile, split into hi / lo f16 planes while they are staged
//            (4 halves = 8 bytes per pixel and plane);
//   weights  all 64 x 7 x 7 x 4 of them as hi / lo planes [cout][ky][8 taps x 4 channels] (tap 7 is zero): one filter ROW is
//            one 32-deep K step, and the eight values a 16x16x32 fragment lane needs (two taps x four channels of one filter
//            row under one output pixel) are 16 contiguous, 16-byte-aligned bytes of the patch: fragments are read straight
//            from the patch with `ds_read_b128`, stride-2 convolution and all.
// Eight waves (one workgroup per CU: 103 KB of LDS), wave w = output row w of the tile: 64 pixels x 64 channels = 16
// accumulator tiles; per filter row 16 fragment reads for 48 MFMAs.  Results leave as float4 stores (a lane's accumulator
// quad is four consecutive channels of one pixel); the per-(tile, channel) sums {S, S^2} of the STORED values go to
// `stats_part` in doubles (one chunk per tile: vfml_instnorm_finalize folds chunks of any shape).
#include "conv_split_common.h"

namespace {

constexpr int ST_TH = 8, ST_TW = 64;                    // output tile
constexpr int ST_PH = 2 * ST_TH + 5;                    // 21 patch rows
constexpr int ST_PW = 2 * ST_TW + 8;                    // 136 patch columns (133 used; a fragment of the last pixel reads 2 more)
constexpr int ST_WP = 232;                              // weight row pitch in halves: 464 B = 116 dwords - the 16 lanes of a
                                                        // fragment read then start in 16 different 4-bank groups
constexpr int ST_K = 7 * 32;                            // halves per weight row in global memory (ky-major, 8 taps x 4 channels)
constexpr int ST_PATCH = ST_PH * ST_PW * 8;             // bytes per plane
constexpr int ST_WBYTES = 64 * ST_WP * 2;               // bytes per plane
constexpr int ST_LDS = 2 * ST_PATCH + 2 * ST_WBYTES;    // 105 088 B

struct StemArgs {
  const f32x4* frames;        // [n][H][W] RGB0
  const _Float16* whi; const _Float16* wlo;       // [64][ST_K]
  const float* bias;          // [64] or null
  float* out;                 // [n][ho][wo][64]
  double* stats_part;         // [n][tiles][64][2] or null
  int n, H, W, ho, wo, tiles_x, tiles_y;
  float w_inv;
};

__global__ __launch_bounds__(512, 2) void stem7x7s2_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* p_hi = smem;
  char* p_lo = smem + ST_PATCH;
  char* w_hi = smem + 2 * ST_PATCH;
  char* w_lo = w_hi + ST_WBYTES;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles = a.tiles_x * a.tiles_y;
  const int img = blockIdx.x / tiles, tl = blockIdx.x - img * tiles;
  const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
  const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;        // input pixel under patch (0, 0)

  // ---- stage the patch (f32 -> hi / lo halves) and the weights -------------------------------------------------
  // (all of a thread's loads are requested before the first one is used: unconditional loads on clamped addresses, the value
  // zeroed afterwards where the patch leaves the image - a load behind a branch makes hipcc wait for each one in turn)
  const f32x4* src = a.frames + (int64_t)img * a.H * a.W;
  constexpr int NPP = (ST_PH * ST_PW + 511) / 512;        // patch pixels per thread (6)
  constexpr int NWC = 64 * (ST_K / 8);                    // 16-byte weight chunks per plane: 28 per row
  constexpr int NWP = (NWC + 511) / 512;                  // ... per thread (4)
  f32x4 pv[NPP];
  bool pok[NPP];
  u32x4 wh[NWP], wl[NWP];
  static_for<NPP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int p = min(t + 512 * q, ST_PH * ST_PW - 1);
    const int py = p / ST_PW, px = p - py * ST_PW;
    const int iy = iy0 + py, ix = ix0 + px;
    pok[q] = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    pv[q] = src[(int64_t)min(max(iy, 0), a.H - 1) * a.W + min(max(ix, 0), a.W - 1)];
  });
  static_for<NWP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int c = min(t + 512 * q, NWC - 1);
    const int row = c / (ST_K / 8), col = c - row * (ST_K / 8);
    wh[q] = *reinterpret_cast<const u32x4*>(a.whi + row * ST_K + col * 8);
    wl[q] = *reinterpret_cast<const u32x4*>(a.wlo + row * ST_K + col * 8);
  });
  static_for<NPP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int p = t + 512 * q;
    if (p < ST_PH * ST_PW) {
      const f32x4 v = pok[q] ? pv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      U8 hi, lo;
      split4(v, hi, lo, 0);
      *reinterpret_cast<uint2*>(p_hi + p * 8) = __builtin_bit_cast(uint2, __builtin_shufflevector(hi.v, hi.v, 0, 1, 2, 3));
      *reinterpret_cast<uint2*>(p_lo + p * 8) = __builtin_bit_cast(uint2, __builtin_shufflevector(lo.v, lo.v, 0, 1, 2, 3));
    }
  });
  static_for<NWP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int c = t + 512 * q;
    if (c < NWC) {
      const int row = c / (ST_K / 8), col = c - row * (ST_K / 8);
      *reinterpret_cast<u32x4*>(w_hi + row * (ST_WP * 2) + col * 16) = wh[q];
      *reinterpret_cast<u32x4*>(w_lo + row * (ST_WP * 2) + col * 16) = wl[q];
    }
  });
  __syncthreads();

  // ---- K loop: one filter row per step, everything resident -----------------------------------------------------
  const int r4 = lane & 15, u4 = lane >> 4;      // 16x16x32: lane -> row lane & 15 of a 16-row tile, 8-element K unit lane >> 4
  // A (pixels): output pixel (row `wave`, column 16 i + r4) under filter row ky reads patch row 2 wave + ky from column
  // 2 (16 i + r4) + 2 u4 on: 8 halves = taps 2 u4, 2 u4 + 1 x 4 channels
  const int aoff = ((2 * wave) * ST_PW + 2 * r4 + 2 * u4) * 8;
  // B (weights): output channel 16 j + r4, K unit u4 of filter row ky
  const int boff = r4 * (ST_WP * 2) + u4 * 16;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    h16x8 bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *reinterpret_cast<const h16x8*>(w_hi + boff + j * 16 * (ST_WP * 2) + ky * 64);
      bl[j] = *reinterpret_cast<const h16x8*>(w_lo + boff + j * 16 * (ST_WP * 2) + ky * 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const h16x8 ah = *reinterpret_cast<const h16x8*>(p_hi + aoff + ky * (ST_PW * 8) + i * (32 * 8));
      const h16x8 al = *reinterpret_cast<const h16x8*>(p_lo + aoff + ky * (ST_PW * 8) + i * (32 * 8));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (weight fragment first: a lane's accumulator quad = four consecutive output channels of one pixel; the order of
        // the three terms is the other kernels')
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al, acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- results: float4 stores; the statistics of what is stored ---------------------------------------------------
  const int oy = oy0 + wave;
  f32x4 bq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bq[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 16 * j + 4 * u4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ox = ox0 + 16 * i + r4;
    const bool ok = oy < a.ho && ox < a.wo;
    float* o = a.out + (((int64_t)img * a.ho + oy) * a.wo + ox) * 64 + 4 * u4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][j][e], a.w_inv, bq[j][e]);
      // (pixels of an edge tile that lie outside the image count as zeros in the sums below)
      acc[i][j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) *reinterpret_cast<f32x4*>(o + 16 * j) = v;
    }
  }
  if (!a.stats_part) return;
  // Per wave: its 64 pixels x 64 channels go through a private 8-KB slab in two halves of 32 channels (the patch and weight
  // images are dead once every wave has left the K loop); lane = (channel, half of the pixels) sums 32 pixels in doubles.
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem) + wave * (64 * 32);
  double* wsum = reinterpret_cast<double*>(smem + 8 * 64 * 32 * 4);       // [8 waves x 2 pixel halves][64 channels][2], behind the slabs
  const int ch = lane & 31, hp = lane >> 5;
#pragma unroll
  for (int hj = 0; hj < 2; ++hj) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        *reinterpret_cast<f32x4*>(&slab[(16 * i + r4) * 32 + 16 * jj + 4 * u4]) = acc[i][2 * hj + jj];
    // (a wave's LDS operations complete in order; the slab is private to the wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int p = 0; p < 32; ++p) {
      // (the first reader of the LDS value is a 32-bit move, then the conversion: no 64-bit-operand op reads an LDS result
      // directly - DESIGN.md section 6)
      const float v = slab[(32 * hp + p) * 32 + ch];
      float vv;
      asm("v_mov_b32 %0, %1" : "=v"(vv) : "v"(v));
      const double d = (double)vv;
      s1 += d;
      s2 += d * d;
    }
    // (the two pixel halves of a channel are folded with the waves below - through LDS, not a cross-lane shuffle)
    wsum[((wave * 2 + hp) * 64 + 32 * hj + ch) * 2 + 0] = s1;
    wsum[((wave * 2 + hp) * 64 + 32 * hj + ch) * 2 + 1] = s2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __syncthreads();
  if (t < 64) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {            // (wave, pixel half) in a fixed order: bitwise reproducible
      s1 += vfml_lds_f64(&wsum[(w * 64 + t) * 2 + 0]);
      s2 += vfml_lds_f64(&wsum[(w * 64 + t) * 2 + 1]);
    }
    double* o = a.stats_part + (((int64_t)img * tiles + tl) * 64 + t) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

}  // namespace

extern "C" int vfml_stem7x7s2_chunks(int h, int w) {
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
  return ((ho + ST_TH - 1) / ST_TH) * ((wo + ST_TW - 1) / ST_TW);
}

extern "C" int vfml_stem7x7s2(const float* frames, int n, int h, int w, const void* w_hi, const void* w_lo, float w_scale,
                              const float* bias, float* out, double* stats_part, void* stream) {
  VFML_REQUIRE(frames && w_hi && w_lo && out, "vfml_stem7x7s2: null pointer");
  VFML_REQUIRE(n > 0 && h >= 1 && w >= 1 && (int64_t)n * h * w < (1ll << 31), "vfml_stem7x7s2: bad n/h/w");
  VFML_REQUIRE(w_scale > 0.f, "vfml_stem7x7s2: bad weight scale");
  VFML_REQUIRE(vfml_aligned16(frames) && vfml_aligned16(w_hi) && vfml_aligned16(w_lo) && vfml_aligned16(out) && vfml_aligned16(bias),
               "vfml_stem7x7s2: operands must be 16-byte aligned");
  StemArgs a;
  a.frames = reinterpret_cast<const f32x4*>(frames);
  a.whi = reinterpret_cast<const _Float16*>(w_hi); a.wlo = reinterpret_cast<const _Float16*>(w_lo);
  a.bias = bias; a.out = out; a.stats_part = stats_part;
  a.n = n; a.H = h; a.W = w; a.ho = (h - 1) / 2 + 1; a.wo = (w - 1) / 2 + 1;
  a.tiles_y = (a.ho + ST_TH - 1) / ST_TH; a.tiles_x = (a.wo + ST_TW - 1) / ST_TW;
  a.w_inv = 1.0f / w_scale;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7x7s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (e != hipSuccess) {
      vfml_set_error("vfml_stem7x7s2: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  const int64_t grid = (int64_t)n * a.tiles_x * a.tiles_y;
  VFML_REQUIRE(grid < (1ll << 31), "vfml_stem7x7s2: too many tiles");
  hipLaunchKernelGGL(stem7x7s2_kernel, dim3((unsigned)grid), dim3(512), ST_LDS, reinterpret_cast<hipStream_t>(stream), a);
  return vfml_check_launch("vfml_stem7x7s2");
}
