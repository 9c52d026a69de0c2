// The encoders' 64 -> 64 channel 3x3 convolutions at half resolution (K2: the four convolutions of `layer1`'s two residual
// blocks, fnet and cnet: eight launches per new frame, 38 GFLOP each) with the instance-norm partial sums of their result.
// On the general shared-stage kernel a 256 x 64 tile runs 18 K steps between a prologue and an epilogue that cost as much as
// the steps: 150 us per 1080p frame alone on the chip (0.30 of the split-f16 ceiling), 200 us beside the iterations.
//
// Here the workgroups are persistent and the WEIGHTS live in registers: wave w computes output channels 16 w .. 16 w + 15 of
// every tile its workgroup takes, and its 36 weight fragments (18 K steps x {hi, lo}) are loaded once per launch.  Per tile
// (4 x 32 output pixels, image width a multiple of 32) the 6 x 34 input pixels under it are staged once (split rows as they
// are in HBM, 272-byte pixel pitch: conflict-free fragment reads); a K step = (32-channel block, tap) reads its activation
// fragments at the tap's shift: no staging loop, no barrier inside the K loop.  The three MFMAs of a product and the order of
// the K steps are those of conv_gemm_tapx_kernel / conv_gemm_dma_kernel (VFML_KORDER_CBLOCK): bit-identical results.  The
// statistics go the same way as there: the tile's stored values pass through LDS and one thread per (row of 32 pixels,
// channel) adds them up in doubles in pixel order (vfml_conv_desc.stats_part, VFML_STATS_ROWS_S16 = 32).
#include "conv_split_common.h"

namespace {

constexpr int EC_C = 64;                               // channels in and out
constexpr int EC_TH = 4, EC_TW = 32;                   // output tile
constexpr int EC_PR = EC_TH + 2, EC_PW = EC_TW + 2;    // input patch
constexpr int EC_PIX = 272;                            // bytes per staged pixel: 64 channels as split rows (256) + 16
constexpr int EC_PATCH = EC_PR * EC_PW * EC_PIX;       // 55 488 B
constexpr int EC_VLD = EC_C + 4;                       // floats per row of the result tile in LDS
constexpr int EC_LDS = EC_PATCH;                       // (the result tile, 128 x 68 floats, reuses the patch)
constexpr int EC_K = 9 * EC_C;                         // 576
constexpr int EC_NS = EC_K / 32;                       // 18 K steps: [cb 2][ky 3][kx 3]
static_assert(EC_TH * EC_TW * EC_VLD * 4 <= EC_PATCH, "the result tile fits where the patch was");

struct EncConvArgs {
  const char* in;              // split rows [n][H][W][ld_in floats], the 64 channels from `in`
  const _Float16* whi; const _Float16* wlo;      // [64][576], K = cb * 288 + (ky * 3 + kx) * 32 + c % 32
  const float* bias;
  float* out; int ldo;         // f32 [n*H*W][ldo]
  double* stats_part;          // [n * H * W / 32][64][2] or null
  int ld_in, n, H, W, tiles_x, tiles_y;
  float w_inv;
};

__global__ __launch_bounds__(256, 2) void enc_conv3x3_c64_kernel(const EncConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sV = reinterpret_cast<float*>(smem);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r4 = lane & 15, u4 = lane >> 4;
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int total = a.n * tiles_img;

  // this wave's weight fragments: output channel 16 wave + r4, K unit u4 of step s (hi and lo planes)
  h16x8 bh[EC_NS], bl[EC_NS];
  static_for<EC_NS>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    bh[s] = *reinterpret_cast<const h16x8*>(a.whi + (16 * wave + r4) * EC_K + s * 32 + u4 * 8);
    bl[s] = *reinterpret_cast<const h16x8*>(a.wlo + (16 * wave + r4) * EC_K + s * 32 + u4 * 8);
  });
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 16 * wave + 4 * u4);

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int img = tile / tiles_img, tl = tile - img * tiles_img;
    const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
    const int oy0 = ty * EC_TH, ox0 = tx * EC_TW;

    // ---- stage the patch: 204 pixels x 16 pieces of 16 bytes (zeros outside the image) ----------------------------------
    constexpr int NPC = EC_PR * EC_PW * 16;              // 3264 pieces
    constexpr int NPT = (NPC + 255) / 256;               // 13 per thread
    // (buffer loads through a per-image descriptor: one 32-bit offset per piece, out-of-image pieces get an offset the
    // descriptor rejects and come back as zeros)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(a.in) + (int64_t)img * a.H * a.W * a.ld_in * 4, 0, a.H * a.W * a.ld_in * 4, 0x00020000);
    u32x4 pv[NPT];
    static_for<NPT>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int c = min(t + 256 * q, NPC - 1);
      const int p = c >> 4, piece = c & 15;
      const int py = p / EC_PW, px = p - py * EC_PW;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      pv[q] = bload16(rin, ok ? ((iy * a.W + ix) * a.ld_in) * 4 + piece * 16 : (int)0x80000000);
    });
    __syncthreads();                                      // (the previous tile's statistics have read the result tile)
    static_for<NPT>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int c = t + 256 * q;
      if (c < NPC) *reinterpret_cast<u32x4*>(smem + (c >> 4) * EC_PIX + (c & 15) * 16) = pv[q];
    });
    __syncthreads();

    // ---- K loop: fragment m = output row m >> 1, columns 16 (m & 1) + r4; step s = (cb, ky, kx) ----------------------------
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* al = smem + r4 * EC_PIX + u4 * 32;
    // (two passes of four fragments: with all eight in one unrolled loop the scheduler keeps more activation fragments in
    // flight than there are registers beside the 144 of the weights)
    static_for<2>([&](auto hc) {
      constexpr int mh = decltype(hc)::value;
      static_for<EC_NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int cb = s / 9, ky = (s % 9) / 3, kx = s % 3;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          const int m = 4 * mh + mm;
          const char* p = al + (((m >> 1) + ky) * EC_PW + 16 * (m & 1) + kx) * EC_PIX + cb * 128;
          const h16x8 ah = *reinterpret_cast<const h16x8*>(p);
          const h16x8 alo = *reinterpret_cast<const h16x8*>(p + 16);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[s], ah, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[s], ah, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[s], alo, acc[m], 0, 0, 0);
        }
        if constexpr (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);     // (a filter row's fragments in flight at most)
      });
      __builtin_amdgcn_sched_barrier(0);
    });

    // ---- result: v = acc * w_inv + bias; f32 rows to HBM, and through LDS for the statistics --------------------------------
    __syncthreads();                                      // every wave is done with the patch
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[m][e] * a.w_inv + bias4[e];
      const int row = m >> 1, col = 16 * (m & 1) + r4;
      const int oy = oy0 + row, ox = ox0 + col;
      *reinterpret_cast<f32x4*>(&sV[(row * EC_TW + col) * EC_VLD + 16 * wave + 4 * u4]) = v;
      if (oy < a.H && ox < a.W)
        *reinterpret_cast<f32x4*>(a.out + (((int64_t)img * a.H + oy) * a.W + ox) * a.ldo + 16 * wave + 4 * u4) = v;
    }
    if (a.stats_part) {
      __syncthreads();
      // one thread per (tile row = 32 consecutive pixels of the image, channel): doubles, in pixel order
      const int row = t >> 6, ch = t & 63;
      const int oy = oy0 + row;
      if (oy < a.H) {
        double s1 = 0.0, s2 = 0.0;
        for (int x = 0; x < EC_TW; ++x) {
          // (the first reader of an LDS result is a 32-bit move, never an op with a 64-bit operand: vfml_common.h, round 2's
          // two-stream finding; tests/test_abi.py scans for it)
          float f = sV[(row * EC_TW + x) * EC_VLD + ch], g;
          asm("v_mov_b32 %0, %1" : "=v"(g) : "v"(f));
          const double v = (double)g;
          s1 += v;
          s2 += v * v;
        }
        const int64_t chunk = (((int64_t)img * a.H + oy) * a.W + ox0) >> 5;
        double* o = a.stats_part + (chunk * EC_C + ch) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
}

}  // namespace

extern "C" int vfml_conv3x3_c64(const float* in, int ld_in, int n, int h, int w, const void* w_hi, const void* w_lo, int kp,
                                float w_scale, const float* bias, float* out, int ldo, double* stats_part, void* stream) {
  VFML_REQUIRE(in && w_hi && w_lo && out && n > 0 && h > 0 && w > 0, "vfml_conv3x3_c64: bad argument");
  VFML_REQUIRE(w % EC_TW == 0, "vfml_conv3x3_c64: the image width must be a multiple of %d (got %d)", EC_TW, w);
  VFML_REQUIRE(kp == EC_K && w_scale > 0.f, "vfml_conv3x3_c64: weight planes [64][%d] (VFML_KORDER_CBLOCK), got row pitch %d", EC_K, kp);
  VFML_REQUIRE((reinterpret_cast<uintptr_t>(in) & 31u) == 0 && ld_in >= EC_C && ld_in % 8 == 0 && vfml_aligned16(w_hi) &&
                   vfml_aligned16(w_lo) && vfml_aligned16(bias) && vfml_aligned16(out) && ldo >= EC_C && ldo % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(stats_part) & 7u) == 0,
               "vfml_conv3x3_c64: a 32-byte aligned split-row input (ld_in %% 8 == 0), 16-byte aligned weights / bias / output "
               "(ldo %% 4 == 0)");
  EncConvArgs a;
  a.in = reinterpret_cast<const char*>(in); a.ld_in = ld_in;
  a.whi = reinterpret_cast<const _Float16*>(w_hi); a.wlo = reinterpret_cast<const _Float16*>(w_lo);
  a.bias = bias; a.out = out; a.ldo = ldo; a.stats_part = stats_part;
  a.n = n; a.H = h; a.W = w; a.tiles_x = w / EC_TW; a.tiles_y = (h + EC_TH - 1) / EC_TH;
  a.w_inv = 1.0f / w_scale;
  const int64_t total = (int64_t)n * a.tiles_x * a.tiles_y;
  VFML_REQUIRE(total < (1ll << 31) && (int64_t)h * w * ld_in * 4 < (1ll << 31), "vfml_conv3x3_c64: an image of the input must stay below 2 GiB");
  const int grid = (int)(total < 512 ? total : 512);     // one workgroup per resident slot (256 CUs x 2)
  hipLaunchKernelGGL(enc_conv3x3_c64_kernel, dim3(grid), dim3(256), EC_LDS, reinterpret_cast<hipStream_t>(stream), a);
  return vfml_check_launch("vfml_conv3x3_c64");
}
