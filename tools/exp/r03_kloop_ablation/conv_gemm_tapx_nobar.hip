// Split-f16 convolution over split-row sources with the activation stage SHARED BY THE TAPS OF A FILTER ROW.
//
// conv_gemm_dma_kernel (conv_gemm_split.hip) stages, per K step = (32-channel block, tap), the 192 activation rows of
// the tile shifted by that tap: a 1x5 convolution pulls every activation byte through LDS-DMA five times, and its
// time is (MFMA time) + (LDS-DMA pieces x their issue cost) - measured, profiles/r01_pmc_dma_conv.md: with the
// activation pieces dropped the same kernel runs 24 % faster.  For a stride-1 "same" convolution the input pixel of
// output pixel m under tap (ky, kx) is the LINEAR index m + (ky - pad_h) W + (kx - pad_w): the kw taps of one filter row
// read the same run of pixels shifted by one.  This kernel stages that run ONCE per (channel block, ky) - the tile's
// TBM pixels plus kw - 1 more - and lets tap kx read its fragments kx rows further down:
//   * LDS-DMA pieces per 192 x 128 K step: 16 (weights) + 25 / kw (activations) instead of 16 + 24;
//   * what the shift breaks - a tap that leaves the image row must read zeros, but the staged neighbour row holds the
//     next image row's pixel - is repaired at the fragment read: a lane whose (row, kx) falls outside [0, W) reads a
//     zero cell instead (one compare and one select per 16-row fragment and step);
//   * taps that leave the image vertically (ky) are whole staged pixels: masked at the DMA, as before.
// LDS: two activation stages of (TBM + 8) x 128 B and ONE weight stage (two of each would be 81 KiB at 192 x 128: one
// KiB more than two workgroups per CU get).  The weight fragments of a step are read into registers first; a second
// barrier then frees the weight stage for the next step's pieces, which land behind this step's MFMAs:
//     [barrier Y: step's weights landed] read B fragments -> [barrier X] -> issue next B (+ a share of the next
//     activation stage) between the first MFMA groups -> MFMAs -> vmcnt(0) -> [barrier Y] ...
// Arithmetic, fragment layouts, bank swizzles, K order (VFML_KORDER_CBLOCK / _CBLOCK64) and the epilogue are those of
// conv_gemm_dma_kernel<.., FASTK = true, .., MF16 = true>: the two kernels are bit-identical on every shape both take
// (tests/test_gpu_kernels.py::test_tap_shared_stage_matches_the_per_tap_kernel).
#include "../../../video-flow-ml_amd/vfml/csrc/conv_split_common.h"

// In-kernel time stamps exist in an EXPERIMENT build only: tools/exp/tapx_stamps.py compiles this file with
// `-include tools/exp/tapx_hooks.h`, which defines the hooks below (per-phase s_memtime sums of one workgroup, an
// s_memrealtime timeline of every workgroup).  In the product they are empty.
#ifndef VFML_TAPX_HOOKS
#define TAPX_HOOK_ENTRY()
#define TAPX_HOOK_KLOOP_BEGIN()
#define TAPX_HOOK_STEP_BEGIN(st)
#define TAPX_HOOK_STAMP(k)
#define TAPX_HOOK_STEP_END()
#define TAPX_HOOK_KLOOP_END(nsteps)
#define TAPX_HOOK_EXIT()
#endif

namespace {

template <int TM, int TN, int WM, int WN, int NM>
__global__ __launch_bounds__(256, 2) void conv_gemm_tapx_kernel(const SplitArgs a) {
  constexpr bool H64 = NM == 5;                       // one MFMA per product over 64-channel steps of hi halves
  constexpr bool BHI = NM == 2 || NM == 1 || H64;     // weight lo slots unused
  constexpr bool AHI = NM == 4 || NM == 1 || H64;     // activation lo slots unused
  constexpr int KSTEP = H64 ? 64 : 32;
  constexpr int NT = 256;
  static_assert(WM * WN == 4, "four waves: two workgroups per CU");
  constexpr int TBM = 32 * TM * WM, TBN = 32 * TN * WN;
  constexpr int NPA = (TBM + VFML_TAPX_KWMAX - 1 + 7) / 8;   // 1-KiB pieces (8 staged pixels x 128 B) of an activation stage
  constexpr int NPI = (NPA + 3) / 4;                         // ... per wave (piece p = 4 i + wave)
  constexpr int BP = TBN / 32;                               // weight pieces per wave and step
  constexpr int ASZ = NPA * 1024, BSZ = TBN * 128;
  constexpr int BOFF = 2 * ASZ, ZOFF = BOFF + BSZ;           // [A0][A1][B][128 B of zeros]
  constexpr int LDC = TBN + 4;
  static_assert(ZOFF % 128 == 0, "the zero cell must keep the fragment address bits 4..6 free");
  constexpr int ASLOTS = (NPI + 1) / 2;                      // issue slots per step for pieces of the next activation stage (kw >= 2)
  static_assert(NPI * 4 <= 64, "activation validity bits: 4 per piece in a register pair");
  static_assert(BP + ASLOTS <= 4 * TM * TN, "one issue slot per MFMA group");

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sC = reinterpret_cast<float*>(smem_raw);
  TAPX_HOOK_ENTRY();

  // this workgroup's tile (XCD x = blockIdx & 7 owns a contiguous share of the tile space)
  const int total = a.mtiles * a.ntiles;
  int tile;
  {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int q = total >> 3, r = total & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile = start + lw;
    if (tile >= start + q + (xcd < r ? 1 : 0)) return;
  }
  const int nt_ = tile % a.ntiles, mt_ = tile / a.ntiles;
  const int m0 = mt_ * TBM, n0 = nt_ * TBN;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t < 32) reinterpret_cast<float*>(smem_raw + ZOFF)[t] = 0.f;

  // ---- loader state -------------------------------------------------------------------------------------------
  // a piece = 8 rows x 128 B; lane -> (row lane >> 3, slot lane & 7); the slot holds source piece slot ^ swizzle(row)
  const int srow = lane >> 3;
  const int sw = (4 * wave + (lane >> 4)) & 7;          // (row >> 1) & 7 of the lane's row in every piece of this wave
  const int pieceL = (lane & 7) ^ (H64 ? sw : swz16(sw));
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in0)) - a.abias, 0, a.bytes0 + a.abias, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wbase), 0, a.bytesb, 0x00020000);
  // activation piece i of this wave = staged pixels 8 (4 i + wave) + srow: byte offset of the lane's 16 bytes for
  // i = 0, tap row 0, channel block 0, relative to the descriptor base (shifted back by abias: never negative)
  int vbaseA = ((m0 + 8 * wave + srow) * a.ld0 + a.d0off) * 4 + (H64 ? pieceL * 32 : (pieceL >> 1) * 32 + (pieceL & 1) * 16);
  if (AHI && !H64 && (pieceL & 1)) vbaseA |= (int)0x80000000;        // lo slots of the activations: never fetched
  // validity of staged pixel (piece i, this lane's row) under tap row ky: bit 4 i + ky SET = outside the image
  // (vertically, or before / past the source, or a row of the stage no tap reads)
  unsigned long long vmask = 0ull;
  {
    const int NH = a.M / a.W;                 // image rows of the whole batch (same convolution: ho = H, wo = W)
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int s = 8 * (4 * i + wave) + srow;
      // under tap row ky the slot holds pixel base + (ky - pad_h) W, base = m0 + s - pad_w; the outputs that read it
      // with a tap inside the image row lie in base's image row qb (output row oy = qb mod H): the tap row is inside
      // the image when 0 <= oy + ky - pad_h < H
      const int base = m0 + s - a.pad_w;
      const int qb = base >= 0 ? base / a.W : -1;
      const int oy = qb - (qb / a.H) * a.H;
      const bool in = 4 * i + wave < NPA && s < TBM + a.kw - 1 && qb >= 0 && qb < NH;
      for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy + ky - a.pad_h;
        if (!(in && iy >= 0 && iy < a.H)) vmask |= 1ull << (4 * i + ky);
      }
    }
  }
  int colbase[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int col = n0 + 32 * j + 8 * wave + srow;
    if constexpr (H64)
      colbase[j] = col < a.cout ? a.whi_off + col * a.Kp * 2 + pieceL * 16 : 0x40000000;
    else
      colbase[j] = col < a.cout && !(BHI && (pieceL & 1)) ? ((pieceL & 1) ? a.wlo_off : a.whi_off) + col * a.Kp * 2 + (pieceL >> 1) * 16
                                                          : 0x40000000;
  }

  // activation piece i of the stage (channel block cbn, tap row kyn) -> buffer nbuf
  auto issue_a = [&](int i, int nbuf, int cbn, int kyn) {
    const int cl = cbn < a.c0 ? cbn : cbn - a.c0;
    const int soff = (kyn * a.W * a.ld0 + cl + i * 32 * a.ld0) * 4 + (cbn < a.c0 ? 0 : a.src1_delta);
    const int bad = (int)((unsigned)(vmask >> (4 * i + kyn)) << 31);                      // sign bit set: not to be fetched
    dma16(r0, bad | vbaseA, soff, smem_raw + nbuf * ASZ + (4 * i + wave) * 1024);
  };
  auto issue_b = [&](int j, int stn) {
    dma16(rb, colbase[j], stn * (H64 ? 128 : 64), smem_raw + BOFF + wave * 1024 + j * 4096);
  };

  // ---- fragment state -----------------------------------------------------------------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int r4 = lane & 15, u4 = lane >> 4;      // 16x16x32: lane -> row lane & 15 of a 16-row tile, 8-channel unit lane >> 4
  const int pb16 = (H64 ? (u4 ^ ((r4 >> 1) & 7)) : ((2 * u4) ^ swz16((r4 >> 1) & 7))) * 16;
  const int boff4 = BOFF + (wn * (32 * TN) + r4) * 128 + pb16;
  constexpr int X1 = H64 ? 64 : 16;              // second fragment of a row: the other 32 channels / the lo halves
  int oxp[2 * TM];                               // output column of the lane's row in 16-row tile i, minus pad_w
#pragma unroll
  for (int i = 0; i < 2 * TM; ++i) {
    const int m = m0 + wm * (32 * TM) + i * 16 + r4;
    oxp[i] = m - (m / a.W) * a.W - a.pad_w;
  }

  f32x4 acc4[2 * TM][2 * TN];
#pragma unroll
  for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc4[i][j][e] = 0.f;

  const int ncb = a.ctot / KSTEP;
  const int nstage = ncb * a.kh;
  const int nsteps = nstage * a.kw;

  // prologue: the whole first stage and the first step's weights
#pragma unroll
  for (int i = 0; i < NPI; ++i)
    if (4 * i + wave < NPA) issue_a(i, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < BP; ++j) issue_b(j, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  TAPX_HOOK_KLOOP_BEGIN();
  int st = 0;
  int cbn = 0, kyn = 0;                 // the stage being loaded (the one after the stage being computed)
  for (int sg = 0; sg < nstage; ++sg) {
    const int abuf = (sg & 1) * ASZ;
    const int nbuf = (sg & 1) ^ 1;
    if (++kyn == a.kh) {
      kyn = 0;
      cbn += KSTEP;
    }
    const bool next_stage = sg + 1 < nstage;
    for (int kx = 0; kx < a.kw; ++kx, ++st) {
      const bool next_step = st + 1 < nsteps;
      TAPX_HOOK_STEP_BEGIN(st);
      // this step's weight fragments into registers, then the weight stage is free for the next step's pieces
      // (the fragments straight from the L2-resident weight planes into registers one step ahead - no weight stage, no
      // barrier X - measured 20-25 % slower: a fragment is 16 rows x 64 bytes, sixteen cache lines per load instruction)
      h16x8 b0[2 * TN], b1[2 * TN];
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) {
        b0[j] = *reinterpret_cast<const h16x8*>(smem_raw + boff4 + j * 2048);
        if constexpr (H64 || !BHI) b1[j] = *reinterpret_cast<const h16x8*>(smem_raw + (boff4 ^ X1) + j * 2048);
      }
      // fragment address of tile row 0 under this tap: staged row r4 + kx (the swizzle follows the shifted row)
      const int R0 = r4 + kx;
      const int swa = (R0 >> 1) & 7;
      const int aoffk = abuf + (wm * (32 * TM) + R0) * 128 + (H64 ? (u4 ^ swa) : ((2 * u4) ^ swz16(swa))) * 16;
      auto a_addr = [&](int i) { return (unsigned)(oxp[i] + kx) < (unsigned)a.W ? aoffk + i * 2048 : ZOFF; };
      h16x8 a0, a1;
      {
        const int ad = a_addr(0);
        a0 = *reinterpret_cast<const h16x8*>(smem_raw + ad);
        if constexpr (H64 || !AHI) a1 = *reinterpret_cast<const h16x8*>(smem_raw + (ad ^ X1));
      }
      asm volatile("s_nop 0" ::: "memory");   // EXPERIMENT: no barrier X
      TAPX_HOOK_STAMP(1);
      __builtin_amdgcn_sched_barrier(0);
      static_for<2 * TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        h16x8 n0_, n1_;
        if constexpr (i + 1 < 2 * TM) {      // next tile row's fragments fly behind this row's MFMAs
          const int ad = a_addr(i + 1);
          n0_ = *reinterpret_cast<const h16x8*>(smem_raw + ad);
          if constexpr (H64 || !AHI) n1_ = *reinterpret_cast<const h16x8*>(smem_raw + (ad ^ X1));
        }
        static_for<2 * TN>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          // (term-major order within the tile row - consecutive MFMAs on different accumulators - measured 2-5 % slower)
          // (the weight fragment is the FIRST operand: a lane's accumulator quad is then four consecutive output channels
          // of one pixel - D[cout 4 (lane >> 4) + e][pixel lane & 15] - which is what the epilogue stores)
          acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[j], a0, acc4[i][j], 0, 0, 0);
          if constexpr (H64) {
            acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a1, acc4[i][j], 0, 0, 0);
          } else {
            if constexpr (!BHI) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[j], a0, acc4[i][j], 0, 0, 0);
            if constexpr (!AHI) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[j], a1, acc4[i][j], 0, 0, 0);
          }
          // behind the first groups: the next step's weight pieces, then this step's share of the next activation stage
          constexpr int g = i * (2 * TN) + j;
          if constexpr (g < BP) {
            if (next_step) issue_b(g, st + 1);
          } else if constexpr (g < BP + ASLOTS) {
            const int ia = kx + (g - BP) * a.kw;
            if (next_stage && ia < NPI && 4 * ia + wave < NPA) issue_a(ia, nbuf, cbn, kyn);
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (i + 1 < 2 * TM) {
          a0 = n0_;
          if constexpr (H64 || !AHI) a1 = n1_;
        }
      });
      TAPX_HOOK_STAMP(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TAPX_HOOK_STAMP(3);
      asm volatile("s_nop 0" ::: "memory");   // EXPERIMENT: no barrier Y
      TAPX_HOOK_STEP_END();
    }
  }

  TAPX_HOOK_KLOOP_END(nsteps);
  // Epilogue in TM slabs through LDS: slab i holds block row i of every wave (WM*32 rows x TBN).
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // (LDS hazards only: a __syncthreads() would also wait for the previous slab's global stores to be acknowledged)
    if (i) lds_barrier();       // every wave is done reading the previous slab
    static_for<2>([&](auto tc) {
      constexpr int t2 = decltype(tc)::value;
      static_for<2 * TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // a lane's quad = pixel row r4 of the 16-row tile, output channels 4 u4 .. 4 u4 + 3 of the 16-column tile
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        static_for<TM>([&](auto ii) { if (decltype(ii)::value == i) v = acc4[2 * decltype(ii)::value + t2][j]; });
        *reinterpret_cast<f32x4*>(&sC[(wm * 32 + t2 * 16 + r4) * LDC + wn * (32 * TN) + j * 16 + 4 * u4]) = v;
      });
    });
    lds_barrier();
    if (!epilogue_rows_fast<TBN, NT>(a, sC, m0, n0, t, WM * 32, 32 * TM, i * 32))
      epilogue_rows<TBN, NT>(a, sC, m0, n0, t, WM * 32, 32 * TM, i * 32);
    if (a.stats_part) {
      // instance-norm statistics of the slab while it is in LDS (vfml_conv_desc.stats_part, as conv_gemm_dma_kernel): the
      // slab's rows wb*32 .. wb*32+31 are the 32 consecutive output pixels from m0 + wb*32*TM + i*32 on; one thread per
      // (block, channel) sums the STORED values (same expression as epilogue_rows) in doubles
      for (int p = t; p < TBN * WM; p += NT) {
        const int ch = p % TBN, wb = p / TBN;
        const int g0 = m0 + wb * (32 * TM) + i * 32;
        if (n0 + ch < a.cout && g0 < a.M) {
          const float b = a.bias ? a.bias[n0 + ch] : 0.f;
          double s1 = 0.0, s2 = 0.0;
          for (int rr = 0; rr < 32; ++rr) {
            if (g0 + rr >= a.M) break;
            const double v = (double)((sC[(wb * 32 + rr) * LDC + ch] * a.w_inv + b) * a.out_scale);
            s1 += v;
            s2 += v * v;
          }
          double* o = a.stats_part + ((int64_t)(g0 >> 5) * a.cout + n0 + ch) * 2;
          o[0] = s1;
          o[1] = s2;
        }
      }
    }
  }
  TAPX_HOOK_EXIT();
}

template <int TM, int TN, int WM, int WN, int NM>
int launch_tapx_k(SplitArgs& a, hipStream_t s) {
  constexpr int TBM = 32 * TM * WM, TBN = 32 * TN * WN;
  constexpr int NPA = (TBM + VFML_TAPX_KWMAX - 1 + 7) / 8;
  constexpr size_t stage = 2 * (size_t)NPA * 1024 + (size_t)TBN * 128 + 128;
  constexpr size_t slab = (size_t)WM * 32 * (TBN + 4) * 4;
  constexpr size_t lds = stage > slab ? stage : slab;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  a.mtiles = (a.M + TBM - 1) / TBM;
  a.ntiles = (a.cout + TBN - 1) / TBN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_tapx_kernel<TM, TN, WM, WN, NM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vfml_set_error("vfml_conv2d_split: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_tapx_kernel<TM, TN, WM, WN, NM>), dim3(a.mtiles * a.ntiles), dim3(256), lds, s, a);
  return vfml_check_launch("vfml_conv2d_split");
}

}  // namespace

// The tile shape (TM TN WM WN as digits) this kernel would run the call on, or 0 when the call is not its: stride-1 "same"
// convolutions over split-row sources on the uniform-step loader, 2..5 taps per filter row, three MFMAs per product or
// one over 64-channel steps.  `cfg` = the shape the per-tap dispatcher chose among its own.
int vfml_detail::tapx_cfg(const SplitArgs& a, int cfg, bool forced) {
  if (!(a.fastk && !a.direct && !a.pointwise && !a.tilebase && a.stride == 1 && a.ho == a.H && a.wo == a.W && a.kw >= 2 &&
        a.kw <= VFML_TAPX_KWMAX && a.kh <= 4 && (a.nm == 3 || a.nm == 5)))
    return 0;
  if (a.cout <= 32) return 0;
  if (forced) return (cfg == 3222 || cfg == 2322 || (cfg == 2241 && a.cout <= 64) || (cfg == 2341 && a.cout <= 96)) ? cfg : 0;   // (VFML_DMA_TILE)
  if (a.cout <= 96) {
    // 256 x 64 / 256 x 96 tiles, when they fill the 512 resident slots of their last round to 85 % (the 1080p 1/8-scale
    // maps are 380 such tiles: three quarters of one round - the per-tap kernel's 128-row tiles serve those better)
    const int64_t tiles = (a.M + 255) / 256, rounds = (tiles + 511) / 512;
    if (tiles * 100 < rounds * 512 * 85) return 0;
    return a.cout <= 64 ? 2241 : 2341;
  }
  return cfg == 3222 || cfg == 2322 ? cfg : 0;
}

int vfml_detail::launch_tapx(SplitArgs& a, int cfg, hipStream_t s) {
  switch (cfg) {
    case 3222: return a.nm == 5 ? launch_tapx_k<3, 2, 2, 2, 5>(a, s) : launch_tapx_k<3, 2, 2, 2, 3>(a, s);
    case 2322: return a.nm == 5 ? launch_tapx_k<2, 3, 2, 2, 5>(a, s) : launch_tapx_k<2, 3, 2, 2, 3>(a, s);
    case 2241: return a.nm == 5 ? launch_tapx_k<2, 2, 4, 1, 5>(a, s) : launch_tapx_k<2, 2, 4, 1, 3>(a, s);
    case 2341: return a.nm == 5 ? launch_tapx_k<2, 3, 4, 1, 5>(a, s) : launch_tapx_k<2, 3, 4, 1, 3>(a, s);
  }
  vfml_set_error("vfml_conv2d_split: no shared-stage variant for tile shape %d", cfg);
  return 1;
}
