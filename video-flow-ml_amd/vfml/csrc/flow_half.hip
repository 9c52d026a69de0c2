// The flow half of the motion encoder as ONE launch: convf1 (7x7 over the 4-channel flow map -> 128 channels, ReLU) and convf2
// (3x3, 128 -> 64 channels, ReLU) - K6's `BasicMotionEncoder.convf1 / convf2` - for the plans that run both layers with one
// MFMA per product (plain f16 operands, f32 accumulate: the shipped mixed plan and the BOF plan).  As three launches
// (vfml_flow_rows7 + two convolutions) the pair cost 77 us of a 1.5-ms iteration for 20 GFLOP: their K axes are 7 and 18
// steps long, so each launch was prologue, epilogue and a 50-MB round trip of the 128-channel map through HBM.
//
// Here a workgroup owns a 4 x 30 tile of output pixels and keeps everything between the flow map and its 64 output channels in
// LDS:
//   patch  the 12 x 38 flow pixels under the tile (f32 -> f16 while staged; zeros outside the image);
//   f1     relu(convf1) on the 6 x 32 pixels the tile's 3x3 windows reach, as f16 rows of 128 channels (ZEROS outside the
//          image: convf2 pads the 128-channel map, not the flow) - the recomputed halo is 1.6x the tile, of a layer that is a
//          third of the pair's work.
// convf1: one filter row = one 32-deep K step (7 taps x 4 channels + 4 zero weights, the vfml_flow_rows7 weight layout), the
// eight values a 16x16x32 fragment lane needs are 16 contiguous bytes of the patch; wave w computes channels 32 w .. 32 w + 31
// with its 14 weight fragments in registers.  convf2: K = [2 blocks of 64 channels][3 x 3 taps][64], two MFMAs per 64-channel
// step - the order of the 64-channel-step kernel (VFML_KORDER_CBLOCK64) - activation fragments read from f1 at the tap's
// shift, wave w computes channels 16 w .. 16 w + 15 with its weight fragments streamed from L2 through a register ring.
// Same products in the same order as the two launches: bit-identical output
// (tests/test_gpu_kernels.py::test_flow_half_kernel_is_the_two_convolutions).
#include "conv_split_common.h"

namespace {

constexpr int FH_TH = 4, FH_TW = 30;                 // output tile
constexpr int FH_FR = FH_TH + 2, FH_FW = 32;         // f1 region: the tile +-1 (32 = two 16-pixel fragments)
constexpr int FH_PR = FH_FR + 6, FH_PW = 40;         // flow patch: the f1 region +-3 (38 columns used; a fragment of the last pixel reads 2 more)
constexpr int FH_PATCH = FH_PR * FH_PW * 8;          // 4 halves per pixel
constexpr int FH_F1PIX = 272;                        // bytes per f1 pixel: 128 halves + 16 (the 16 lanes of a fragment start in 16 different 4-bank groups)
constexpr int FH_F1 = FH_FR * FH_FW * FH_F1PIX;
constexpr int FH_LDS = FH_PATCH + FH_F1;             // 56 064 B: two workgroups per CU
constexpr int FH_K1 = 7 * 32, FH_K2 = 9 * 128;
constexpr int FH_RING = 12;                          // convf2 weight fragments in flight per wave

struct FlowHalfArgs {
  const f32x4* flow;           // [n][h][w] flow quads (fwd x, y, bwd x, y)
  const _Float16* w1;          // [128][FH_K1] hi plane, K = ky * 32 + kx * 4 + c
  const _Float16* w2;          // [64][FH_K2] hi plane, K = cb * 576 + (ky * 3 + kx) * 64 + c
  const float* b1; const float* b2;
  float* out; int ld_out;      // split rows: channel 0 of the 64 at out[p * ld_out]
  int n, h, w, tiles_x, tiles_y;
  float w1_inv, w2_inv;
};

__global__ __launch_bounds__(256, 2) void flow_half_kernel(const FlowHalfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* f1 = smem + FH_PATCH;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles = a.tiles_x * a.tiles_y;
  const int img = blockIdx.x / tiles, tl = blockIdx.x - img * tiles;
  const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
  const int oy0 = ty * FH_TH, ox0 = tx * FH_TW;
  const int r4 = lane & 15, u4 = lane >> 4;          // 16x16x32: row lane & 15 of a 16-row tile, 8-element K unit lane >> 4

  // ---- stage the flow patch (f32 -> f16) -------------------------------------------------------------------------
  const f32x4* src = a.flow + (int64_t)img * a.h * a.w;
  constexpr int NPP = (FH_PR * FH_PW + 255) / 256;
  f32x4 pv[NPP];
  bool pok[NPP];
  static_for<NPP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int p = min(t + 256 * q, FH_PR * FH_PW - 1);
    const int py = p / FH_PW, px = p - py * FH_PW;
    const int iy = oy0 - 4 + py, ix = ox0 - 4 + px;
    pok[q] = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
    pv[q] = src[(int64_t)min(max(iy, 0), a.h - 1) * a.w + min(max(ix, 0), a.w - 1)];
  });
  // convf1 weight fragments of this wave: channels 32 wave + 16 j + r4, K unit u4 of filter row ky
  h16x8 w1f[7][2];
  static_for<7>([&](auto kc) {
    constexpr int ky = decltype(kc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      w1f[ky][j] = *reinterpret_cast<const h16x8*>(a.w1 + (32 * wave + 16 * j + r4) * FH_K1 + ky * 32 + u4 * 8);
  });
  f32x4 b1v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) b1v[j] = *reinterpret_cast<const f32x4*>(a.b1 + 32 * wave + 16 * j + 4 * u4);
  static_for<NPP>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int p = t + 256 * q;
    if (p < FH_PR * FH_PW) {
      const f32x4 v = pok[q] ? pv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      const h16x2 h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
      *reinterpret_cast<uint2*>(patch + p * 8) = uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    }
  });
  __syncthreads();

  // ---- convf1 on the 6 x 32 region -> f1 (f16, zeros outside the image) -------------------------------------------
  // fragment m: region row m >> 1, columns 16 (m & 1) + r4; under filter row ky a lane reads patch row (m >> 1) + ky from
  // column 16 (m & 1) + r4 + 2 u4 on: 8 halves = taps 2 u4, 2 u4 + 1 x 4 channels (8-byte aligned: two 8-byte reads)
  static_for<FH_FR>([&](auto rc) {
    constexpr int fy = decltype(rc)::value;
    f32x4 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<7>([&](auto kc) {
      constexpr int ky = decltype(kc)::value;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* p = patch + (((fy + ky) * FH_PW) + 16 * s + r4 + 2 * u4) * 8;
        const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
        const h16x8 av = __builtin_bit_cast(h16x8, u32x4{lo.x, lo.y, hi.x, hi.y});
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[s][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[ky][j], av, acc[s][j], 0, 0, 0);
      }
    });
    // a lane's quad: channels 32 wave + 16 j + 4 u4 .. + 3 of pixel (fy, 16 s + r4)
    const int iy = oy0 - 1 + fy;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ix = ox0 - 1 + 16 * s + r4;
      const bool inside = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(acc[s][j][e] * a.w1_inv + b1v[j][e], 0.f) : 0.f;
        const h16x2 h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<uint2*>(f1 + (fy * FH_FW + 16 * s + r4) * FH_F1PIX + (32 * wave + 16 * j + 4 * u4) * 2) =
            uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
      }
    }
  });
  __syncthreads();

  // ---- convf2 on the 4 x 30 tile: wave = 16 output channels, 8 pixel fragments (row m >> 1, columns (m & 1 ? 14 : 0) + r4) ----
  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const _Float16* w2row = a.w2 + (16 * wave + r4) * FH_K2 + u4 * 8;
  constexpr int NS = FH_K2 / 32;                       // 36 K steps of 32: [cb 2][tap 9][half 2]
  h16x8 ring[FH_RING];
  static_for<FH_RING>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    ring[s] = *reinterpret_cast<const h16x8*>(w2row + (s >> 1) * 64 + (s & 1) * 32);
  });
  // fragment base of pixel fragment m in f1: region row m >> 1 (+ ky), column (m & 1 ? 14 : 0) + r4 (+ kx)
  const char* f1l = f1 + r4 * FH_F1PIX + u4 * 16;
  static_for<NS>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int cb = s / 18, tap = (s % 18) / 2, half = s & 1, ky = tap / 3, kx = tap % 3;
    const h16x8 bv = ring[s % FH_RING];
    if constexpr (s + FH_RING < NS) {
      constexpr int n = s + FH_RING;
      ring[s % FH_RING] = *reinterpret_cast<const h16x8*>(w2row + (n >> 1) * 64 + (n & 1) * 32);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const h16x8 av = *reinterpret_cast<const h16x8*>(f1l + (((m >> 1) + ky) * FH_FW + ((m & 1) ? 14 : 0) + kx) * FH_F1PIX +
                                                        (cb * 64 + half * 32) * 2);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bv, av, acc[m], 0, 0, 0);
    }
  });

  // ---- epilogue: relu, split rows (a lane's quad: channels 16 wave + 4 u4 .. + 3 = quad u4 & 1 of unit 2 wave + (u4 >> 1)) ----
  const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + 16 * wave + 4 * u4);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int oy = oy0 + (m >> 1), col = ((m & 1) ? 14 : 0) + r4;
    const int ox = ox0 + col;
    if (oy >= a.h || ox >= a.w || ((m & 1) && r4 < 2)) continue;       // (columns 14, 15 belong to the first fragment)
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[m][e] * a.w2_inv + b2v[e], 0.f);
    vfml_h16x2 h0, h1, l0, l1;
    vfml_split2(v[0], v[1], h0, l0);
    vfml_split2(v[2], v[3], h1, l1);
    char* u = reinterpret_cast<char*>(a.out + ((int64_t)img * a.h * a.w + (int64_t)oy * a.w + ox) * a.ld_out + 8 * (2 * wave + (u4 >> 1))) +
              (u4 & 1) * 8;
    *reinterpret_cast<uint2*>(u) = uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    *reinterpret_cast<uint2*>(u + 16) = uint2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
  }
}

}  // namespace

extern "C" int vfml_flow_half(const float* flow, int n, int h, int w, const void* w1_hi, int kp1, float w1_scale, const float* b1,
                              const void* w2_hi, int kp2, float w2_scale, const float* b2, float* out, int ld_out, void* stream) {
  VFML_REQUIRE(flow && w1_hi && w2_hi && b1 && b2 && out && n > 0 && h > 0 && w > 0, "vfml_flow_half: bad argument");
  VFML_REQUIRE(kp1 == FH_K1 && kp2 == FH_K2, "vfml_flow_half: weight planes [128][%d] (vfml_flow_rows7 layout) and [64][%d] "
               "(VFML_KORDER_CBLOCK64), got row pitches %d / %d", FH_K1, FH_K2, kp1, kp2);
  VFML_REQUIRE(w1_scale > 0.f && w2_scale > 0.f, "vfml_flow_half: weight scales must be positive");
  VFML_REQUIRE(vfml_aligned16(flow) && vfml_aligned16(w1_hi) && vfml_aligned16(w2_hi) && vfml_aligned16(b1) && vfml_aligned16(b2) &&
                   (reinterpret_cast<uintptr_t>(out) & 31u) == 0 && ld_out >= 64 && ld_out % 8 == 0,
               "vfml_flow_half: 16-byte aligned operands, a 32-byte aligned split-row output with ld_out %% 8 == 0");
  FlowHalfArgs a;
  a.flow = reinterpret_cast<const f32x4*>(flow);
  a.w1 = reinterpret_cast<const _Float16*>(w1_hi); a.w2 = reinterpret_cast<const _Float16*>(w2_hi);
  a.b1 = b1; a.b2 = b2; a.out = out; a.ld_out = ld_out;
  a.n = n; a.h = h; a.w = w;
  a.tiles_x = (w + FH_TW - 1) / FH_TW; a.tiles_y = (h + FH_TH - 1) / FH_TH;
  a.w1_inv = 1.0f / w1_scale; a.w2_inv = 1.0f / w2_scale;
  const int64_t grid = (int64_t)n * a.tiles_x * a.tiles_y;
  VFML_REQUIRE(grid < (1ll << 31), "vfml_flow_half: too many tiles");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&flow_half_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
    if (e != hipSuccess) {
      vfml_set_error("vfml_flow_half: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(flow_half_kernel, dim3((unsigned)grid), dim3(256), FH_LDS, reinterpret_cast<hipStream_t>(stream), a);
  return vfml_check_launch("vfml_flow_half");
}
