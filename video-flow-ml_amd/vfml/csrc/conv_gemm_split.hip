// Implicit-GEMM convolution / NT-GEMM on the CDNA4 f16 matrix cores with fp32-grade accuracy.
//
// "split-f16": every operand x is carried as two halves, hi = f16(x) and lo = f16(x - hi), and
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi                 (a_lo*b_lo ~ 2^-22 |ab| is dropped)
// f16 x f16 products are exact in the f32 accumulator, so the result carries ~22 mantissa bits,
// at 3 v_mfma_f32_32x32x16_f16 (1024 FLOP/clk/SIMD each) per product instead of one
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD): 16/3 = 5.3x the f32 MFMA rate, one accumulator.
// lo is an f16 subnormal when |x| < 2^-3 (absolute error <= 2^-25 there; gfx950's MFMA takes f16
// subnormals unflushed - measured).  The pre-split operand (weights, correlation targets) is
// scaled by a power of two at split time so that its lo halves are normal numbers; the epilogue
// multiplies the accumulator by the inverse (exact).
//
// Two kernels share the arithmetic and the epilogue:
//  * conv_gemm_dma_kernel   - sources in the split-row format (VFML_FMT_S16: per pixel and 8-channel
//    group 16 B of hi halves then 16 B of lo halves), staged into LDS by `buffer_load ... lds` (LDS-DMA):
//    every update-block convolution, the correlation GEMMs, the MemFlow attention GEMMs.  Described
//    in front of the kernel.
//  * conv_gemm_split_kernel - fp32 NHWC sources (encoders, the 4-channel flow convolution), split
//    while register-staged into LDS.  Tiling: 128 pixels x BN channels per 256-thread workgroup, K
//    stepped by 32; LDS image per operand plane [k/8][row] 16-byte units (8 halves), row stride padded
//    by 2 units (staging writes cover all 32 banks, a wave's fragment read is 32 consecutive units =
//    one MFMA operand, conflict-free).  Two register staging sets: the global loads of K step k+2 are
//    issued before the MFMAs of step k, the landed loads of step k+1 are split and written to the other
//    LDS buffer after them; one barrier per K step.  Loads are unconditional (padding taps / K tails
//    read out-of-range buffer offsets = zeros) so that no load sits behind a divergent branch.
// Weights (and the pooled target features of the correlation GEMM) are pre-split once by
// vfml_split_f16 into two f16 planes [cout][Kp], zero padded.
// Epilogue: the accumulator tile is transposed through LDS and written as float4 rows (bias,
// activation and the GRU gate math applied on the way).
#include "conv_split_common.h"

namespace {

// BIGC: every source row has >= BK channels, so a K step never spans more than two taps and the
// (channel, tap) state advances without divisions.  !BIGC (4-channel stem / flow convs) recomputes
// it by division each step.
// IN16: the sources are already in the split-row format (VFML_FMT_S16: per pixel and 8-channel group
// 16 B of hi halves then 16 B of lo halves), so the two 16-byte loads of a unit ARE its hi and lo
// LDS images and no conversion happens in the loop.
// NM: which terms of the split product are formed (SplitArgs::nm): 3 = a_hi w_hi + a_hi w_lo + a_lo w_hi;
// 2 = the weight operand as plain f16 (a_hi w_hi + a_lo w_hi); 4 = the activation operand as plain f16
// (a_hi w_hi + a_hi w_lo); 1 = both operands plain f16 (a_hi w_hi).
template <int BN, int WM, int WN, bool BIGC, bool IN16, int NM = 3>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_split_kernel(const SplitArgs a) {
  constexpr bool ALO = NM == 3 || NM == 2, BLO = NM == 3 || NM == 4;
  constexpr int NT = WM * WN * 64;  // threads: 256 (4 waves) or 512 (8 waves, finer MFMA interleave per SIMD)
  constexpr int LR = NT / 4;        // rows covered by one pass of the loader (4 k-groups per row)
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int AU = BM / LR;   // A units per thread per K step (each unit: 8 k of one row)
  constexpr int BU = BN / LR;   // B units per thread per K step; 0 -> only threads with lrow < BN load B
  constexpr int BUN = BU > 0 ? BU : 1;
  constexpr int RSA = BM + 2;   // unit row stride per k-group (padded)
  constexpr int RSB = BN + 2;
  constexpr int SA = KG * RSA;  // units per A plane per buffer
  constexpr int SB = KG * RSB;
  constexpr int LDC = BN + 4;   // epilogue tile row stride (floats)

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  h16x8* sAh = reinterpret_cast<h16x8*>(smem_raw);
  h16x8* sAl = sAh + 2 * SA;
  h16x8* sBh = sAl + 2 * SA;
  h16x8* sBl = sBh + 2 * SB;
  float* sC = reinterpret_cast<float*>(smem_raw);

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
  const int lrow = t >> 2;  // 0..LR-1

  // Per-row gather state, fixed for the whole K loop.  All A offsets are 32-bit element offsets
  // (the host checks that every source spans < 2^31 floats).
  // tapok[i] bit (ky*kw+kx) = that tap of row i lies inside the image (kh*kw <= 64, host check)
  int rp0[AU], rp1[AU];
  unsigned long long tapok[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int m = m0 + lrow + LR * i;
    int iy0[1], ix0[1];   // (kept as arrays of one to reuse the expressions below)
    tapok[i] = 0ull;
    if (m < a.M) {
      const int hw = a.ho * a.wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / a.wo;
      const int ox = rem - oy * a.wo;
      iy0[0] = oy * a.stride - a.pad_h;
      ix0[0] = ox * a.stride - a.pad_w;
      const int pix = (n * a.H + iy0[0]) * a.W + ix0[0];
      rp0[i] = pix * a.ld0 + a.d0off;
      rp1[i] = pix * a.ld1 + a.d1off;
      for (int ky = 0; ky < a.kh; ++ky)
        for (int kx = 0; kx < a.kw; ++kx)
          if ((unsigned)(iy0[0] + ky) < (unsigned)a.H && (unsigned)(ix0[0] + kx) < (unsigned)a.W)
            tapok[i] |= 1ull << (ky * a.kw + kx);
    } else {
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
  // Byte offset of this thread's B rows; rows past cout start beyond the descriptor (planes are
  // < 1 GiB, host check), so adding the k offset keeps them out of range: zeros, no select.
  int colbase[BUN];
#pragma unroll
  for (int i = 0; i < BUN; ++i) {
    const int col = n0 + lrow + LR * i;
    const bool colok = (BU > 0 || lrow < BN) && col < a.cout;
    colbase[i] = colok ? col * a.Kp * 2 : 0x40000000;
  }

  struct Stage {
    f32x4 a[AU][2];
    h16x8 bh[BUN], bl[BUN];
  };
  Stage st0, st1;

  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.whi), 0, a.bytesw, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wlo), 0, a.bytesw, 0x00020000);

  int offs16[AU];
  // Issues the loads of the K step starting at k0 (must be called in increasing k0 order).
  auto load_tile = [&](Stage& s, int k0) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (IN16 && hf == 1) {   // second 16 bytes of the same unit: the lo halves
#pragma unroll
        for (int i = 0; i < AU; ++i) s.a[i][1] = __builtin_bit_cast(f32x4, bload16(r0, offs16[i]));
        break;
      }
      const bool kok = k0 + kg * 8 + hf * 4 < a.K;
      int c = kc[hf];
      const int ky = kky[hf], kx = kkx[hf];
      const bool s1 = c >= a.c0;
      const int ld = s1 ? a.ld1 : a.ld0;
      if (s1) c -= a.c0;
      const int tapoff = (ky * a.W + kx) * ld + c;
      const int tap = kok ? ky * a.kw + kx : 63;   // k >= K: tap >= kh*kw, bit never set (host: kh*kw < 64)
#pragma unroll
      for (int i = 0; i < AU; ++i) {
        const bool ok = (tapok[i] >> tap) & 1ull;
        const int off = ((s1 ? rp1[i] : rp0[i]) + tapoff) * 4;
        s.a[i][hf] = __builtin_bit_cast(f32x4, bload16(r0, ok ? off : OOB));
        if (IN16) offs16[i] = ok ? off + 16 : OOB;
      }
      if (BIGC) {
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
    // Kp is a multiple of BK and the planes are zero padded, so every k of a K step < nk exists;
    // the one prefetch past the end (k0 = nk*BK) lands in the next row or beyond the descriptor and
    // is never consumed.
    const int kb2 = (k0 + kg * 8) * 2;
#pragma unroll
    for (int i = 0; i < BUN; ++i) {
      s.bh[i] = __builtin_bit_cast(h16x8, bload16(rh, colbase[i] + kb2));
      s.bl[i] = __builtin_bit_cast(h16x8, bload16(rl, colbase[i] + kb2));
    }
  };
  auto store_tile = [&](const Stage& s, int buf) {
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      U8 hi, lo;
      if (IN16) {
        hi.v = __builtin_bit_cast(h16x8, s.a[i][0]);
        lo.v = __builtin_bit_cast(h16x8, s.a[i][1]);
      } else {
        split4(s.a[i][0], hi, lo, 0);
        split4(s.a[i][1], hi, lo, 4);
      }
      sAh[buf * SA + kg * RSA + lrow + LR * i] = hi.v;
      sAl[buf * SA + kg * RSA + lrow + LR * i] = lo.v;
    }
#pragma unroll
    for (int i = 0; i < BUN; ++i) {
      if (BU > 0 || lrow < BN) {
        sBh[buf * SB + kg * RSB + lrow + LR * i] = s.bh[i];
        sBl[buf * SB + kg * RSB + lrow + LR * i] = s.bl[i];
      }
    }
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

  auto compute = [&](int buf) {
    const int oa = buf * SA + wm * (BM / WM) + r;
    const int ob = buf * SB + wn * (BN / WN) + r;
#pragma unroll
    for (int ks = 0; ks < KG / 2; ++ks) {
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = sAh[oa + (2 * ks + half) * RSA + i * 32];
        if constexpr (ALO) al[i] = sAl[oa + (2 * ks + half) * RSA + i * 32];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = sBh[ob + (2 * ks + half) * RSB + j * 32];
        if constexpr (BLO) bl[j] = sBl[ob + (2 * ks + half) * RSB + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (BLO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          if constexpr (ALO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // Two staging sets, two K steps of prefetch: the loads of step k+2 are issued before the MFMAs of
  // step k; the loads of step k+1 (issued one iteration earlier) are split and written to the other
  // LDS buffer after them.  Loads past K are descriptor-out-of-range (zeros, no traffic), so the
  // loop body has no conditionals (a conditional load makes the compiler's s_waitcnt placement
  // assume the not-taken path and drain the fresh loads too) and the step count is rounded up to
  // even (one all-zero step when odd) so that the unrolled pair has a single exit.
  const int nk = ((a.K + BK - 1) / BK + 1) & ~1;
  load_tile(st0, 0);
  load_tile(st1, BK);
  store_tile(st0, 0);
  __syncthreads();
  // (round 1, timing only: without the in-loop loads and LDS writes the MFMA + fragment-read phase alone ran at 500-570
  // TFLOP/s algorithmic against 285-310 with staging - profiles/HISTORY.md "what bounds the conv kernel")
  for (int kt = 0; kt < nk; kt += 2) {
    load_tile(st0, (kt + 2) * BK);
    __builtin_amdgcn_sched_barrier(0);   // loads first ...
    compute(0);
    __builtin_amdgcn_sched_barrier(0);   // ... their consumers (split + LDS write) only after the MFMAs
    store_tile(st1, 1);
    __syncthreads();
    load_tile(st1, (kt + 3) * BK);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    store_tile(st0, 0);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS tile -> float4 rows ------------------------------------
  // (every wave passed the loop's final barrier, so the staging buffers are free)
  acc_to_lds<TM, TN, LDC>(acc, sC, wm * (BM / WM), wn * (BN / WN), r, half);
  __syncthreads();
  if (!epilogue_rows_fast<BN, NT>(a, sC, m0, n0, t, BM, 32, 0)) epilogue_rows<BN, NT>(a, sC, m0, n0, t);
  if (a.stats_part) {
    // instance-norm statistics of the tile while it is in LDS: thread = (channel, row group), doubles like the
    // stand-alone pass; the value is the stored one (same expression as epilogue_rows, no addend / activation here)
    constexpr int RG = NT / BN;            // row groups
    const int ch = t % BN, rg = t / BN;
    double s1 = 0.0, s2 = 0.0;
    if (n0 + ch < a.cout) {
      const float b = a.bias ? a.bias[n0 + ch] : 0.f;
      for (int row = rg; row < BM; row += RG) {
        if (m0 + row >= a.M) break;
        const double v = (double)((sC[row * LDC + ch] * a.w_inv + b) * a.out_scale);
        s1 += v;
        s2 += v * v;
      }
    }
    __syncthreads();                       // every thread is done with the tile: reuse it for the fold
    double* fold = reinterpret_cast<double*>(sC);
    fold[(rg * BN + ch) * 2] = s1;
    fold[(rg * BN + ch) * 2 + 1] = s2;
    __syncthreads();
    if (rg == 0 && n0 + ch < a.cout) {
      for (int g = 1; g < RG; ++g) {
        s1 += vfml_lds_f64(&fold[(g * BN + ch) * 2]);
        s2 += vfml_lds_f64(&fold[(g * BN + ch) * 2 + 1]);
      }
      double* o = a.stats_part + ((int64_t)(m0 / BM) * a.cout + n0 + ch) * 2;
      o[0] = s1;
      o[1] = s2;
    }
  }
}

// ---- LDS-DMA variant (split-row sources, 128 x BN tile) ------------------------------------------
// The staging path of the kernel above (buffer_load -> VGPR -> ds_write_b128) costs the LDS store
// path 13 cycles per wave-instruction and 64 cache lines per load instruction (one row per lane).
// Here every K step of a row is one 128-byte line of the split-row source - four 8-channel units,
// each 16 B of hi halves then 16 B of lo halves - and one `buffer_load_dwordx4 ... lds` moves eight
// rows x 128 B (eight full lines) straight into LDS: no staging registers, no ds_write, 1/8 of the
// lines per instruction.  The LDS image is row-major, 128 B per row, and since an LDS-DMA writes lane l
// at base + 16*l, the bank swizzle is applied on the SOURCE side: the lane that fills slot (row r,
// 16-byte piece s) fetches piece s ^ ((r >> 1) & 7) of that row.  A fragment read of 32 consecutive
// rows at one logical piece is then conflict-free for ds_read_b128's four 16-lane groups.
// The weight operand uses the same image: piece 2u is unit u of the hi plane, 2u+1 of the lo plane
// (both planes lie in one descriptor window).
// Two LDS stages; the DMAs of step k+1 are issued right after the barrier that opens step k and are
// waited for (vmcnt(0)) before the next one, so they have a whole step of MFMAs to land.

// Waves WM x WN, each 32*TM x 32*TN outputs; 4 waves run two workgroups per CU, 8 waves one.
// The grid is persistent: a workgroup walks tiles start + lw, start + lw + nl, ... of its XCD's
// contiguous share of the tile space, and issues the first K step of the next tile in the slot where
// the last step of the current one has nothing left to prefetch - the next tile's first loads are
// in flight during the epilogue (which matters when K is short: the correlation GEMM has 8 steps).
// PERSIST (plain wide f32 outputs = the correlation GEMMs): persistent grid, accumulators stored
// straight to global, the next tile's first loads and this tile's stores overlap the neighbours' MFMAs.
// FASTK: the uniform-step loader (SplitArgs::fastk) as a compile-time choice - with both loaders in one
// body hipcc stops unrolling the MFMA loops of the larger tiles and the accumulators go to scratch.
// CSWAP: VFML_CONV_SWAP_CROSS as a compile-time choice (GEMM form only; a runtime branch in the MFMA loop spills).
// NM: terms of the split product (SplitArgs::nm).  2: the weight operand is taken as plain f16 - its lo slots are never
// fetched (those lanes of a weight piece carry an out-of-range offset) nor read, a product is a_hi b + a_lo b; the
// operand may then be ONE f16 plane without a lo plane at all (SplitArgs::bhi).  4: the ACTIVATION operand as plain
// f16 instead (its lo slots not fetched: a_hi b_hi + a_hi b_lo).  1: both, one MFMA per product.
// MF16: the products run on v_mfma_f32_16x16x32_f16 instead of 32x32x16 (same FLOPs per cycle, same LDS bytes per
// FLOP: a fragment is 16 rows x all 32 channels of the step instead of 32 rows x 16 channels).  The chip holds a higher
// clock on the 16x16 shape under an MFMA-dense load (MI355X_MICROARCH.md, DVFS give-back item 7).  The lane -> (row,
// piece) map of a fragment read differs, so the bank swizzle of the LDS image does too (swz16 below); the
// accumulators are 16 x 16 tiles (4 registers each).  Not built for the persistent GEMM form.

template <int TM, int TN, int WM, int WN, bool PERSIST, bool FASTK, bool CSWAP = false, int NM = 3, bool MF16 = false, bool H16 = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void conv_gemm_dma_kernel(const SplitArgs a) {
  static_assert(!H16 || PERSIST, "VFML_FMT_F16 outputs (H16) exist in the persistent GEMM form only");
  static_assert(!(MF16 && PERSIST), "the persistent GEMM form runs 32x32x16 MFMAs (its 16x16x32 epilogue branches below are not maintained)");
  // NM == 5 (H64): one MFMA per product like NM == 1, and a K step covers 64 channels of hi halves only: a staged row's 128
  // bytes are the hi halves of eight 8-channel units (fetched at a 32-byte stride from the split-row source; the weight
  // row's are contiguous in its hi plane) - no lane fetches a lo half, half the steps, barriers and LDS-DMA
  // instructions of NM == 1 for the same MFMAs.  Uniform-step loader only, channel counts multiples of 64, weights in
  // VFML_KORDER_CBLOCK64 order (or 1x1).  The image's swizzle is the plain one for both MFMA shapes (a fragment's two
  // row groups differ by ONE piece here, not two: brute-force checked conflict-free).
  constexpr bool H64 = NM == 5;
  static_assert(!H64 || FASTK, "64-channel steps exist for the uniform-step loader only");
  constexpr bool BHI = NM == 2 || NM == 1 || H64;     // weight lo slots unused
  constexpr bool AHI = NM == 4 || NM == 1 || H64;     // activation lo slots unused
  constexpr int KSTEP = H64 ? 64 : BK;               // channels per K step
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int TBM = 32 * TM * WM, TBN = 32 * TN * WN;   // workgroup tile
  constexpr int AP = TBM / (8 * NW), BP = TBN / (8 * NW);  // 1-KiB pieces (8 rows x 128 B) per wave per K step
  static_assert(TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0, "tile rows must split into whole pieces per wave");
  constexpr int ASZ = TBM * 128, BSZ = TBN * 128, STG = ASZ + BSZ;
  constexpr int LDC = TBN + 4;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sC = reinterpret_cast<float*>(smem_raw);

  // this workgroup's tiles (XCD x = blockIdx & 7 owns a contiguous share of the tile space)
  // (a.ksplit == 2, persistent GEMM form: every tile is two work items, one per half of the K axis)
  const int total = a.mtiles * a.ntiles * (PERSIST ? a.ksplit : 1);
  int tile, tile_end, tile_step;
  {
    const int G = gridDim.x, xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int q = total >> 3, r = total & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile_step = (G - xcd + 7) >> 3;
    tile = start + lw;
    tile_end = start + q + (xcd < r ? 1 : 0);
  }
  if (tile >= tile_end) return;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // loader: piece j of this wave covers tile rows 8*NW*j + 8*wave .. +7; lane -> (row lane>>3, slot lane&7)
  const int lrow = 8 * wave + (lane >> 3);
  // slot ^ ((row >> 1) & 7)  (32x32x16 fragments), slot ^ swz16((row >> 1) & 7)  (16x16x32 fragments)
  const int piece = (lane & 7) ^ (MF16 && !H64 ? swz16((4 * wave + (lane >> 4)) & 7) : ((4 * wave + (lane >> 4)) & 7));
  const int kg = piece >> 1, hl = piece & 1;

  int rp0[AP], rp1[AP];              // byte offsets of the row's first tap in source 0 / 1
  unsigned long long tapok[AP];
  int colbase[BP];
  int kc, kky, kkx;   // channel within the tap and tap position of k = k0 + 8*kg, advanced by BK per step
  int scb = 0, sky = 0, skx = 0;   // uniform-step loader: channel block and tap of the step (scalars)
  int m0, n0;
  int ks = 0;          // a.ksplit == 2: the half of the K axis this work item covers
  __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in0)) - a.abias, 0, a.bytes0 + a.abias, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wbase), 0, a.bytesb, 0x00020000);

  // steps per work item (rounded up to even; a second half that reaches past K reads zeros on the source side)
  const int nk = PERSIST && a.ksplit == 2 ? (((a.Kp / KSTEP + 1) >> 1) + 1) & ~1 : (a.Kp / KSTEP + 1) & ~1;
  auto setup = [&](int tl) {
    int nt, mt;
    if (PERSIST && a.ksplit == 2) {
      ks = tl & 1;
      tl >>= 1;
    }
    if (a.ntiles >= 8) {
      // wide outputs (GEMMs): the 64 tiles an XCD has in flight form an 8 x 8 block, so every operand
      // tile it pulls into its L2 serves 8 workgroups (n-fastest order streams the whole second operand
      // once per row tile, with no reuse when it exceeds the 4 MiB L2)
      constexpr int GM = 8;
      const int gsz = GM * a.ntiles;
      const int g = tl / gsz, rem = tl - g * gsz;
      const int left = a.mtiles - g * GM;
      const int mrows = left < GM ? left : GM;
      nt = rem / mrows;
      mt = g * GM + (rem - nt * mrows);
    } else {
      nt = tl % a.ntiles;
      mt = tl / a.ntiles;
    }
    m0 = mt * TBM;
    n0 = nt * TBN;
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int m = m0 + 8 * NW * j + lrow;
      tapok[j] = 0ull;
      rp0[j] = rp1[j] = 0;
      if (a.pointwise) {       // 1x1, stride 1, no padding: output pixel m IS input pixel m (GEMM rows)
        if (m < a.M) {
          tapok[j] = 1ull;
          const int mrow = a.tilebase ? m - m0 : m;      // tilebase: offsets from the tile's first row
          rp0[j] = (mrow * a.ld0 + a.d0off) * 4;
          rp1[j] = (mrow * a.ld1 + a.d1off) * 4;
        }
      } else if (m < a.M) {
        const int hw = a.ho * a.wo;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int oy = rem / a.wo;
        const int ox = rem - oy * a.wo;
        const int iy0 = oy * a.stride - a.pad_h, ix0 = ox * a.stride - a.pad_w;
        const int pix = (n * a.H + iy0) * a.W + ix0;
        rp0[j] = (pix * a.ld0 + a.d0off) * 4;
        rp1[j] = (pix * a.ld1 + a.d1off) * 4;
        for (int ky = 0; ky < a.kh; ++ky)
          for (int kx = 0; kx < a.kw; ++kx)
            if ((unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W)
              tapok[j] |= 1ull << (ky * a.kw + kx);
      }
    }
    {
      const int k = kg * 8;
      const int tap = k / a.ctot;
      kc = k - tap * a.ctot;
      kky = tap / a.kw;
      kkx = tap - kky * a.kw;
    }
    if (FASTK) {
      // uniform-step loader: rp0 = row base + this lane's bytes within a 32-channel block, shifted by abias so
      // that it is never negative (the descriptor base is shifted back); tapok holds the INVERTED tap mask
#pragma unroll
      for (int j = 0; j < AP; ++j) {
        rp0[j] += a.abias + (H64 ? piece * 32 : kg * 32 + hl * 16);
        tapok[j] = ~tapok[j];
        if (AHI && !H64 && hl) rp0[j] |= (int)0x80000000;      // lo slots of the activations: never fetched
      }
      scb = ks * nk * KSTEP;         // (ksplit: pointwise calls only - the K axis is the channel axis)
      sky = skx = 0;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int col = n0 + 8 * NW * j + lrow;
      if constexpr (H64)
        colbase[j] = col < a.cout ? a.whi_off + col * a.Kp * 2 + piece * 16 : (a.bhi ? 0x7ffffff0 : 0x40000000);
      else
        colbase[j] = col < a.cout && !(BHI && hl) ? (hl ? a.wlo_off : a.whi_off) + col * a.Kp * 2 + kg * 16 : (a.bhi ? 0x7ffffff0 : 0x40000000);
    }
    if (a.tilebase) {
      // GEMM rows of a source that can exceed what one descriptor spans: base it at this tile's first row
      const int rows = a.M - m0 < TBM ? a.M - m0 : TBM;
      r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0) + (int64_t)m0 * a.ld0, 0,
                                             ((rows - 1) * a.ld0 + a.d0off + a.c0) * 4, 0x00020000);
    }
  };

  // The loads of a K step: offsets first (prep_step, VALU), then one LDS-DMA instruction per piece
  // (issue_piece), which the step loop places BETWEEN the MFMA groups of the step before - a wave issues in
  // order, and a piece takes ~100 cycles to get through the texture path when the CU is filling LDS
  // at its rate; issued in a block ahead of the MFMAs, the pieces of a step hold the wave's matrix pipe idle.
  int va[AP];
  int soffA = 0, soffB = 0;      // scalar offsets of the step (buffer soffset operand)
  auto prep_step = [&](int k0) {
    soffB = (k0 + ks * nk * BK) * (H64 ? 4 : 2);   // k0 counts steps x 32: a 64-channel step is 128 bytes of the hi plane
    if constexpr (FASTK) {
      // every lane of the step reads the same tap of the same 32-channel block: the tap / channel / source
      // offset is one scalar, the per-piece work is "row valid for this tap?" -> two VALU instructions
      const int cl = scb < a.c0 ? scb : scb - a.c0;
      soffA = ((sky * a.W + skx) * a.ld0 + cl) * 4 + (scb < a.c0 ? 0 : a.src1_delta);
      const unsigned stap = sky * a.kw + skx;
      // the rounding-up step of an odd step count lies past the last channel block: zeros, like every K tail
      const int past = scb >= a.ctot ? (int)0x80000000 : 0;
#pragma unroll
      for (int j = 0; j < AP; ++j) {
        const int bad = __builtin_amdgcn_sbfe((int)(unsigned)tapok[j], stap, 1u);   // -1: tap outside the image (or row past M)
        va[j] = (bad & (int)0x80000000) | past | rp0[j];
      }
      if (++skx == a.kw) {
        skx = 0;
        if (++sky == a.kh) {
          sky = 0;
          scb += KSTEP;
        }
      }
      return;
    }
    if constexpr (!FASTK) {
    soffA = 0;
    // channel-block order: this unit's channel can lie in the zero padding of the last block;
    // tap order: the K tail of the last step
    const bool kok = a.korder ? kc < a.ctot : k0 + kg * 8 < a.K;
    int c = kc;
    const bool s1 = c >= a.c0;
    const int ld = s1 ? a.ld1 : a.ld0;
    if (s1) c -= a.c0;
    const int tapoff = ((kky * a.W + kkx) * ld + c) * 4 + hl * 16;
    const int tap = kok ? kky * a.kw + kkx : 63;   // bit 63 is never set (kh*kw < 64, host check)
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const bool ok = ((tapok[j] >> tap) & 1ull) && !(AHI && hl);
      va[j] = ok ? (s1 ? rp1[j] : rp0[j]) + tapoff : OOB;
    }
    if (a.korder) {          // next tap of the same 32 channels; after the last tap the next 32 channels
      if (++kkx == a.kw) {
        kkx = 0;
        if (++kky == a.kh) {
          kky = 0;
          kc += BK;
        }
      }
    } else {                 // next 32 channels of the same tap; after the last channel the next tap
      int cn = kc + BK;
      if (cn >= a.ctot) {
        cn -= a.ctot;
        if (++kkx == a.kw) {
          kkx = 0;
          ++kky;
        }
      }
      kc = cn;
    }
    }
  };
  auto issue_piece = [&](int stg, int pc) {
    if (pc < AP)
      dma16(r0, va[pc < AP ? pc : 0], soffA, smem_raw + stg * STG + wave * (8 * 128) + pc * (8 * NW * 128));
    else if (pc < AP + BP)
      dma16(rb, colbase[pc >= AP && pc < AP + BP ? pc - AP : 0], soffB,
            smem_raw + stg * STG + ASZ + wave * (8 * 128) + (pc - AP) * (8 * NW * 128));
  };
  auto issue_all = [&](int stg) {
#pragma unroll
    for (int pc = 0; pc < AP + BP; ++pc) issue_piece(stg, pc);
  };

  const int wm = wave / WN;
  const int wn = wave - wm * WN;
  const int r = lane & 31;
  const int half = lane >> 5;
  // fragment address of logical piece x at row r: row*128 + ((x ^ ((r>>1)&7)) * 16); x = 4*ks + 2*half + hl
  const int q16 = ((((r >> 1) & 7) ^ (H64 ? half : 2 * half)) * 16);
  const int aoff = (wm * (32 * TM) + r) * 128 + q16;
  const int boff = ASZ + (wn * (32 * TN) + r) * 128 + q16;
  // 16x16x32: lane -> row lane & 15 of a 16-row tile, 8-channel unit lane >> 4 (hi piece 2u, lo piece 2u + 1)
  const int r4 = lane & 15, u4 = lane >> 4;
  const int p16 = (H64 ? (u4 ^ ((r4 >> 1) & 7)) : ((2 * u4) ^ swz16((r4 >> 1) & 7))) * 16;
  const int aoff4 = (wm * (32 * TM) + r4) * 128 + p16;
  const int boff4 = ASZ + (wn * (32 * TN) + r4) * 128 + p16;

  f32x16 acc[MF16 ? 1 : TM][MF16 ? 1 : TN];
  f32x4 acc4[MF16 ? 2 * TM : 1][MF16 ? 2 * TN : 1];

  // MFMAs of the step in stage `stg`; after each (i, j) group of three, one piece of the step that
  // prep_step prepared goes out to stage `lstg` (when `issue`)
  auto compute = [&](int stg, int lstg, bool issue) {
    const char* base = smem_raw + stg * STG;
    if constexpr (MF16) {
      // one MFMA covers 32 channels of the step (H64: two of them cover its 64): 2*TN weight fragments stay in registers,
      // the 2*TM activation tiles stream through; after each of the first half of the (i, j) groups one piece of the next
      // step goes out
      constexpr int KS = H64 ? 2 : 1;
      static_for<KS>([&](auto sc) {
        constexpr int ks = decltype(sc)::value;
        h16x8 bh[2 * TN], bl[2 * TN];
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
          bh[j] = *reinterpret_cast<const h16x8*>(base + (boff4 ^ (ks * 64)) + j * 2048);
          if constexpr (!BHI) bl[j] = *reinterpret_cast<const h16x8*>(base + (boff4 ^ 16) + j * 2048);
        }
        static_for<2 * TM>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          h16x8 ah, al;
          ah = *reinterpret_cast<const h16x8*>(base + (aoff4 ^ (ks * 64)) + i * 2048);
          if constexpr (!AHI) al = *reinterpret_cast<const h16x8*>(base + (aoff4 ^ 16) + i * 2048);
          static_for<2 * TN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            // (the weight fragment is the FIRST operand: a lane's accumulator quad is then four consecutive output
            // channels of one pixel - D[cout 4 (lane >> 4) + e][pixel lane & 15] - which is what the epilogue stores)
            acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah, acc4[i][j], 0, 0, 0);
            if constexpr (!BHI) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah, acc4[i][j], 0, 0, 0);
            if constexpr (!AHI) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al, acc4[i][j], 0, 0, 0);
            // the next step's pieces go out behind the first HALF of this step's MFMA groups (behind a quarter, or all of
            // them: +-1 %, round 2)
            constexpr int GROUPS4 = KS * 4 * TM * TN / 2;
            constexpr int g = (ks * 2 * TM + i) * (2 * TN) + j;
            constexpr int PER4 = (AP + BP + GROUPS4 - 1) / GROUPS4;
            if constexpr (g < GROUPS4) {
              if (issue) {
                static_for<PER4>([&](auto qc) { issue_piece(lstg, g * PER4 + decltype(qc)::value); });
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      });
      return;
    }
    if constexpr (H64) {
      // 32x32x16, hi halves only: four 16-channel sub-steps per 64-channel step, logical piece 2 ks + half
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        h16x8 ah[TM], bh[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const h16x8*>(base + (aoff ^ (ks * 32)) + i * 4096);
#pragma unroll
        for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const h16x8*>(base + (boff ^ (ks * 32)) + j * 4096);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            constexpr int GROUPS = 2 * TM * TN;     // the first half of the step's 4*TM*TN groups
            const int g = (ks * TM + i) * TN + j;
            constexpr int PER = (AP + BP + GROUPS - 1) / GROUPS;
            if (issue && g < GROUPS) {
#pragma unroll
              for (int q = 0; q < PER; ++q) issue_piece(lstg, g * PER + q);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const h16x8*>(base + (aoff ^ (ks * 64)) + i * 4096);
        if constexpr (!AHI) al[i] = *reinterpret_cast<const h16x8*>(base + (aoff ^ (ks * 64 + 16)) + i * 4096);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const h16x8*>(base + (boff ^ (ks * 64)) + j * 4096);
        if constexpr (!BHI) bl[j] = *reinterpret_cast<const h16x8*>(base + (boff ^ (ks * 64 + 16)) + j * 4096);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (CSWAP) {         // VFML_CONV_SWAP_CROSS (correlation GEMMs of the backward problems)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          } else {
            if constexpr (!BHI) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            if constexpr (!AHI) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          }
          constexpr int GROUPS = TM * TN;      // the first half of the step: the second half covers the pieces' L2 latency
          const int g = (ks * TM + i) * TN + j;
          // spread AP+BP pieces over the first GROUPS groups (the first groups get one more when it does not divide)
          constexpr int PER = (AP + BP + GROUPS - 1) / GROUPS;
          if (issue && g < GROUPS) {
#pragma unroll
            for (int q = 0; q < PER; ++q) issue_piece(lstg, g * PER + q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  // The step count is rounded up to even (one all-zero step when odd: out of range on the source side,
  // the weight side reads the next row's first step - multiplied by zeros).
  constexpr int NSTORE = TM * TN * 4;                    // direct epilogue: 16-byte stores per thread, all issued
  constexpr int RELAXED = NSTORE < 63 ? NSTORE : 63;     // vmcnt that still covers the older DMAs
  constexpr int NSTORE_T = NSTORE + TM * TN * 4;         // with the transposed second output
  constexpr int RELAXED_T = NSTORE_T < 63 ? NSTORE_T : 63;
  bool stores_behind = false;   // PERSIST: the previous tile's stores are still in flight behind this tile's first DMAs
  auto step_pair = [&](int kt, bool last, int next) {
    // A tile's first wait must not drain the previous tile's stores: vmcnt counts in issue order and the
    // first step's DMAs were issued BEFORE them, so leaving min(63, NSTORE) operations outstanding still
    // waits for every DMA (the stores then have one whole K step to finish before the next vmcnt(0)).
    if (PERSIST && stores_behind) {
      if (a.out_t) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RELAXED_T) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RELAXED) : "memory");
      stores_behind = false;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                       // stage 0 landed for every wave; stage 1's readers are done
    prep_step((kt + 1) * BK);
    __builtin_amdgcn_sched_barrier(0);
    compute(0, 1, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool issue = true;
    if (!last) {
      prep_step((kt + 2) * BK);
    } else if (PERSIST && next < tile_end) {
      setup(next);                         // the next tile's first step flies during this tile's epilogue
      prep_step(0);
    } else {
      issue = false;
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(1, 0, issue);
  };

  setup(tile);
  prep_step(0);
  issue_all(0);
  while (true) {
    if constexpr (MF16) {
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc4[i][j][e] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const int cur_m0 = m0, cur_n0 = n0, cur_ks = ks;
    const int next = PERSIST ? tile + tile_step : tile_end;
    // (direct epilogue) this tile's bias quad, loaded before the K loop: a load in the epilogue would make its
    // s_waitcnt drain the previous tile's stores as well
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (PERSIST && a.bias) {
      const int bc = cur_n0 + wn * (32 * TN) + (lane % (8 * TN)) * 4;
      if (bc < a.cout) bv = *reinterpret_cast<const f32x4*>(a.bias + bc);
    }
    // (projection epilogue: the tile's slice of the second weight is requested here and waits in 24 registers for the K
    // loop's stages to become free - requested behind the K loop, its round trip was exposed once per tile)
    constexpr bool PROJ = !PERSIST && MF16 && FASTK && (NM == 3 || NM == 4) && TBN == 128 && WM == 2 && NW == 4;   // vfml_conv_desc.proj_out
    u32x4 pw[PROJ_PIECES];      // (dead where PROJ is false)
    if constexpr (PROJ) {
      if (a.proj_out) proj_weights_load(a, cur_n0, t, pw);
    }
    for (int kt = 0; kt < nk - 2; kt += 2) step_pair(kt, false, next);
    step_pair(nk - 2, true, next);

    if constexpr (PERSIST) {
      // Wide plain-f32 outputs (the correlation GEMM: K is short, the tile's 4 bytes per product dominate).
      // Each wave transposes its own 32*TM x 32*TN block through a private 32 x 32*TN slab in stage 1
      // (stage 0 is already receiving the next tile) and writes whole rows of it as 16-byte stores: an
      // instruction covers 64/(8*TN) rows x 128*TN contiguous bytes.  (Dword stores straight from the
      // accumulators reach 2.4 TB/s, a third of what the chip writes with 16 bytes per lane.)
      // Buffer stores through a per-tile descriptor: rows past M fall outside it, lanes past cout get an
      // out-of-range offset, so every store instruction is issued (the relaxed vmcnt counts on NSTORE).
      const int rows_valid = a.M - cur_m0 < TBM ? a.M - cur_m0 : TBM;
      const int cols_valid = a.cout - cur_n0 < TBN ? a.cout - cur_n0 : TBN;
      // (out_h16: VFML_FMT_F16 outputs - one f16 per element, 8-byte stores of four; same store count, half the bytes)
      constexpr int ES = H16 ? 2 : 4;
      constexpr int STORE_NT = 2;            // buffer-store aux bits: nt - the volume is streamed out once
      // (ksplit: the second half's partial sums go to the workspace; the host adds them after the launch)
      char* tbase = reinterpret_cast<char*>(cur_ks ? a.out_k1 : a.out) + ((int64_t)cur_m0 * a.ldo + cur_n0) * ES;
      const __amdgpu_buffer_rsrc_t ro =
          __builtin_amdgcn_make_buffer_rsrc(tbase, 0, ((rows_valid - 1) * a.ldo + cols_valid) * ES, 0x00020000);
      constexpr int WC = 32 * TN;            // slab row, floats
      constexpr int L4 = WC / 4;             // lanes per slab row
      constexpr int RPI = 64 / L4;           // rows per store instruction
      float* ws = reinterpret_cast<float*>(smem_raw + STG) + wave * (32 * WC);
      const int c4 = (lane % L4) * 4, rr = lane / L4;
      const int gcol = wn * WC + c4;                                   // column within the tile
      const int lbase = gcol < cols_valid ? ((wm * (32 * TM) + rr) * a.ldo + gcol) * ES : OOB;
      __syncthreads();                       // every wave is done reading stage 1
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (MF16) {
          static_for<2>([&](auto tc) {
            constexpr int t2 = decltype(tc)::value;
            static_for<2 * TN>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = 0.f;
                static_for<TM>([&](auto ii) { if (decltype(ii)::value == i) v = acc4[2 * decltype(ii)::value + t2][j][e]; });
                ws[(t2 * 16 + 4 * u4 + e) * WC + j * 16 + r4] = v;
              }
            });
          });
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) ws[((e & 3) + 8 * (e >> 2) + 4 * half) * WC + j * 32 + r] = acc[i][j][e];
        }
        // (same wave, LDS operations complete in order: no barrier between the writes and the reads)
#pragma unroll
        for (int p = 0; p < 32 / RPI; ++p) {
          f32x4 v = *reinterpret_cast<const f32x4*>(ws + (p * RPI + rr) * WC + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = (v[e] * a.w_inv + bv[e]) * a.out_scale;   // same expression as epilogue_rows
            if (a.epilogue == VFML_EPI_RELU) v[e] = fmaxf(v[e], 0.f);
          }
          const int roff = (i * 32 + p * RPI) * a.ldo * ES;
          if constexpr (H16) {
            const h16x2 p0 = {(_Float16)v[0], (_Float16)v[1]}, p1 = {(_Float16)v[2], (_Float16)v[3]};
            const u32x2 hv = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
            __builtin_amdgcn_raw_buffer_store_b64(hv, ro, gcol < cols_valid ? lbase + roff : OOB, 0, STORE_NT);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, gcol < cols_valid ? lbase + roff : OOB, 0,
                                                   STORE_NT);
          }
        }
        if constexpr (PERSIST && TN == 2 && FASTK) if (a.out_t) {   // (the general-loader instantiation would spill)
          // The transposed copy, out_t[column][row]: the block goes through the same 8 KiB as [column][row]
          // with the row index rotated by (column >> 1) - the 32 lanes of a store group (one row, 32
          // columns) then hit 32 different banks - and leaves as 16-byte runs along the rows: a store
          // instruction covers 8 rows of out_t x 128 contiguous bytes.
          if constexpr (MF16) {
            static_for<2>([&](auto tc) {
              constexpr int t2 = decltype(tc)::value;
              static_for<2 * TN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float v = 0.f;
                  static_for<TM>([&](auto ii) { if (decltype(ii)::value == i) v = acc4[2 * decltype(ii)::value + t2][j][e]; });
                  const int row = t2 * 16 + 4 * u4 + e, col = j * 16 + r4;
                  ws[col * 32 + ((row + (col >> 1)) & 31)] = v;
                }
              });
            });
          } else {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = (e & 3) + 8 * (e >> 2) + 4 * half, col = j * 32 + r;
              ws[col * 32 + ((row + (col >> 1)) & 31)] = acc[i][j][e];
            }
          }
          const int rows_t = cols_valid, cols_t = rows_valid;      // extent of the transposed tile
          char* tbase_t = reinterpret_cast<char*>(cur_ks ? a.out_t_k1 : a.out_t) + ((int64_t)cur_n0 * a.ld_out_t + cur_m0) * ES;
          const __amdgpu_buffer_rsrc_t rt =
              __builtin_amdgcn_make_buffer_rsrc(tbase_t, 0, ((rows_t - 1) * a.ld_out_t + cols_t) * ES, 0x00020000);
          const int g = lane >> 3, cl = lane & 7;
#pragma unroll
          for (int pp = 0; pp < 8; ++pp) {
            const int c = pp * 8 + cl;                              // column of the wave's block = row of out_t
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ws[c * 32 + ((4 * g + k + (c >> 1)) & 31)] * a.w_inv * a.out_scale;
            const int q0 = wm * (32 * TM) + i * 32 + 4 * g;         // first of the four pixels (columns of out_t)
            const int trow = wn * WC + c;
            const bool ok = trow < rows_t && q0 < cols_t;           // M % 4 == 0 (host check): a quad is whole
            if constexpr (H16) {
              const h16x2 p0 = {(_Float16)v[0], (_Float16)v[1]}, p1 = {(_Float16)v[2], (_Float16)v[3]};
              const u32x2 hv = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
              __builtin_amdgcn_raw_buffer_store_b64(hv, rt, ok ? (trow * a.ld_out_t + q0) * 2 : OOB, 0, STORE_NT);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rt,
                                                     ok ? (trow * a.ld_out_t + q0) * 4 : OOB, 0, STORE_NT);
            }
          }
        }
      }
      stores_behind = true;
    } else {
      // Epilogue in TM slabs: slab i holds block row i of every wave (WM*32 rows x TBN) in LDS.  (Straight from the
      // accumulators - a lane holds four consecutive channels of a pixel - with 8-byte loads and stores, no LDS: measured
      // 0.7 ms per 1080p field slower than the rows below.)
      {
      static_for<TM>([&](auto slab) {       // (static_for, not `#pragma unroll`: the body is large and a slab loop left rolled
        constexpr int i = decltype(slab)::value;   // would index the accumulators dynamically - scratch)
        __syncthreads();     // every wave is done with the stage buffers / with the previous slab
        if constexpr (PROJ && i == 0) {
          if (a.proj_out) proj_weights_store(pw, smem_raw + proj_lds_off<TBN>(), t);
        }
        if constexpr (MF16) {
          // 16 x 16 tiles: a lane's quad = pixel row lane & 15, output channels 4 (lane >> 4) .. + 3; block row i = tile
          // rows 2i, 2i + 1 (i is the index of the enclosing slab loop: a compile-time constant through static_for)
          static_for<2>([&](auto tc) {
            constexpr int t2 = decltype(tc)::value;
            static_for<2 * TN>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              f32x4 v = {0.f, 0.f, 0.f, 0.f};
              static_for<TM>([&](auto ii) { if (decltype(ii)::value == i) v = acc4[2 * decltype(ii)::value + t2][j]; });
              *reinterpret_cast<f32x4*>(&sC[(wm * 32 + t2 * 16 + r4) * LDC + wn * (32 * TN) + j * 16 + 4 * u4]) = v;
            });
          });
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const int col = wn * (32 * TN) + j * 32 + r;
            sC[row * LDC + col] = acc[i][j][e];
          }
        }
        __syncthreads();
        bool projected = false;
        if constexpr (PROJ) {
          if (a.proj_out) {      // (host: ReLU, no addend; the 256-channel map itself is not stored)
            epilogue_proj_slab<TBN, NT, NM == 4>(a, sC, cur_m0, cur_n0, t, WM * 32, 32 * TM, i * 32);
            projected = true;
          }
        }
        // (what the fast rows take, the 16x16x32 forms have already written straight from their accumulators)
        if (!projected && !epilogue_rows_fast<TBN, NT>(a, sC, cur_m0, cur_n0, t, WM * 32, 32 * TM, i * 32))
          epilogue_rows<TBN, NT>(a, sC, cur_m0, cur_n0, t, WM * 32, 32 * TM, i * 32);
        if (a.stats_part) {
          // instance-norm statistics of the slab while it is in LDS (vfml_conv_desc.stats_part, split-row sources): the
          // slab's rows wb*32 .. wb*32+31 are the 32 consecutive output pixels from cur_m0 + wb*32*TM + i*32 on; one
          // thread per (block, channel) sums the STORED values (same expression as epilogue_rows) in doubles
          for (int p = t; p < TBN * WM; p += NT) {
            const int ch = p % TBN, wb = p / TBN;
            const int g0 = cur_m0 + wb * (32 * TM) + i * 32;
            if (cur_n0 + ch < a.cout && g0 < a.M) {
              const float b = a.bias ? a.bias[cur_n0 + ch] : 0.f;
              double s1 = 0.0, s2 = 0.0;
              for (int rr = 0; rr < 32; ++rr) {
                if (g0 + rr >= a.M) break;
                const double v = (double)((sC[(wb * 32 + rr) * LDC + ch] * a.w_inv + b) * a.out_scale);
                s1 += v;
                s2 += v * v;
              }
              double* o = a.stats_part + ((int64_t)(g0 >> 5) * a.cout + cur_n0 + ch) * 2;
              o[0] = s1;
              o[1] = s2;
            }
          }
        }
      });
      }
    }
    if (!PERSIST || next >= tile_end) break;
    tile = next;
  }
}


template <int TM, int TN, int WM, int WN, bool PERSIST, bool FASTK, bool CSWAP = false, int NM = 3, bool MF16 = false, bool H16 = false>
int launch_dma_k(SplitArgs& a, hipStream_t s) {
  constexpr int TBM = 32 * TM * WM, TBN = 32 * TN * WN;
  constexpr size_t stage = 2 * (size_t)(TBM + TBN) * 128;
  constexpr size_t slab = (size_t)WM * 32 * (TBN + 4) * 4;
  constexpr size_t lds = (PERSIST || stage > slab) ? stage : slab;
  static_assert(lds <= 160 * 1024, "LDS");
  a.mtiles = (a.M + TBM - 1) / TBM;
  a.ntiles = (a.cout + TBN - 1) / TBN;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_dma_kernel<TM, TN, WM, WN, PERSIST, FASTK, CSWAP, NM, MF16, H16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vfml_set_error("vfml_conv2d_split: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  // PERSIST: one workgroup per resident slot (256 CUs x 2 or 1), fewer when there are fewer tiles
  const int64_t total = (int64_t)a.mtiles * a.ntiles * (PERSIST ? a.ksplit : 1);     // work items
  const int64_t slots = 256 * (WM * WN == 4 ? 2 : 1);
  const int grid = (int)(PERSIST && total > slots ? slots : total);
  hipLaunchKernelGGL((conv_gemm_dma_kernel<TM, TN, WM, WN, PERSIST, FASTK, CSWAP, NM, MF16, H16>), dim3(grid), dim3(WM * WN * 64), lds, s, a);
  return vfml_check_launch("vfml_conv2d_split");
}

template <int TM, int TN, int WM, int WN>
int launch_dma(SplitArgs& a, hipStream_t s) {
  if (a.direct) {   // (128 x 128: the one persistent tile shape that does not spill)
    // The GEMM forms stay on 32x32x16: measured with 16x16x32 (an experiment since removed) the 32400^2 volume gains 4 %, the
    // MemFlow read-out 2 %, the 1080p field nothing - and v_mfma_f32_16x16x32_f16 is NOT symmetric in its operands to the
    // last bit (a volume stored transposed and the reverse problem computed directly differ in the last ulp, which the
    // 32x32x16 form never does: tests/test_gpu_kernels.py::test_wide_gemm_with_transposed_second_output), so the
    // sliding job's "volume + transposed volume from one pass" would stop being bit-identical to from-scratch fields.
    if (a.out_h16) {      // VFML_FMT_F16 outputs (host: implies fastk)
      if (a.cswap) return launch_dma_k<2, 2, 2, 2, true, true, true, 3, false, true>(a, s);
      if (a.nm == 2) return launch_dma_k<2, 2, 2, 2, true, true, false, 2, false, true>(a, s);
      if (a.nm == 1) return launch_dma_k<2, 2, 2, 2, true, true, false, 1, false, true>(a, s);
      if (a.nm == 5) return launch_dma_k<2, 2, 2, 2, true, true, false, 5, false, true>(a, s);
      a.nm = 3;
      return launch_dma_k<2, 2, 2, 2, true, true, false, 3, false, true>(a, s);
    }
    if (a.cswap) return launch_dma_k<2, 2, 2, 2, true, true, true>(a, s);     // (host: cswap implies fastk and nm == 3)
    if (a.fastk && a.nm == 2) return launch_dma_k<2, 2, 2, 2, true, true, false, 2>(a, s);  // (host: bhi implies fastk, nm <= 2)
    if (a.fastk && a.nm == 1) return launch_dma_k<2, 2, 2, 2, true, true, false, 1>(a, s);
    if (a.fastk && a.nm == 5) return launch_dma_k<2, 2, 2, 2, true, true, false, 5>(a, s);
    if (a.nm == 2 && a.bhi) { vfml_set_error("vfml_conv2d_split: a weight operand without lo plane needs the uniform-step GEMM form"); return 1; }
    a.nm = 3;       // (the general-loader GEMM form exists at full precision only: never less accurate than asked)
    return a.fastk ? launch_dma_k<2, 2, 2, 2, true, true>(a, s) : launch_dma_k<2, 2, 2, 2, true, false>(a, s);
  }
  if constexpr (WM * WN == 4) {   // the shapes the dispatcher picks by itself: 16x16x32 MFMAs (MF16)
    // VFML_MF32=1: the 32x32x16 shape for the full-precision uniform-step variants (A/B; MF16 is 9-12 % faster on the
    // 1080p update-block shapes: the chip holds a higher clock on it)
    static const int mf32 = getenv("VFML_MF32") ? atoi(getenv("VFML_MF32")) : 0;
    if constexpr (TM * TN >= 2) {
      if (a.fastk) {
        if (mf32 && a.nm == 3) return launch_dma_k<TM, TN, WM, WN, false, true>(a, s);
        if (a.nm == 2) return launch_dma_k<TM, TN, WM, WN, false, true, false, 2, true>(a, s);
        if (a.nm == 4) return launch_dma_k<TM, TN, WM, WN, false, true, false, 4, true>(a, s);
        if (a.nm == 1) return launch_dma_k<TM, TN, WM, WN, false, true, false, 1, true>(a, s);
        if (a.nm == 5) return launch_dma_k<TM, TN, WM, WN, false, true, false, 5, true>(a, s);
        return launch_dma_k<TM, TN, WM, WN, false, true, false, 3, true>(a, s);
      }
    }
    if (a.nm == 5) {     // (cannot happen: the host picks 64-channel steps only where a uniform-step variant exists)
      vfml_set_error("vfml_conv2d_split: no 64-channel-step variant for this tile shape");
      return 1;
    }
    a.fastk = 0; a.abias = 0;
    if (a.nm == 2) return launch_dma_k<TM, TN, WM, WN, false, false, false, 2, true>(a, s);
    if (a.nm == 4) return launch_dma_k<TM, TN, WM, WN, false, false, false, 4, true>(a, s);
    if (a.nm == 1) return launch_dma_k<TM, TN, WM, WN, false, false, false, 1, true>(a, s);
    return launch_dma_k<TM, TN, WM, WN, false, false, false, 3, true>(a, s);
  }
  vfml_set_error("vfml_conv2d_split: no such tile shape");
  return 1;
}

template <int BN, int WM, int WN, bool BIGC, bool IN16, int NM>
int launch_nm(const SplitArgs& a, hipStream_t s) {
  constexpr size_t stage = 2 * 2 * (KG * (BM + 2) + KG * (BN + 2)) * 16;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * 4;
  constexpr size_t lds = stage > ctile ? stage : ctile;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_split_kernel<BN, WM, WN, BIGC, IN16, NM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      vfml_set_error("vfml_conv2d_split: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return 2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_gemm_split_kernel<BN, WM, WN, BIGC, IN16, NM>), dim3(a.mtiles * a.ntiles), dim3(WM * WN * 64), lds, s, a);
  return vfml_check_launch("vfml_conv2d_split");
}

template <int BN, int WM, int WN, bool BIGC, bool IN16>
int launch(const SplitArgs& a, hipStream_t s) {
  if (a.nm == 1) return launch_nm<BN, WM, WN, BIGC, IN16, 1>(a, s);
  if (a.nm == 2) return launch_nm<BN, WM, WN, BIGC, IN16, 2>(a, s);
  if (a.nm == 4) return launch_nm<BN, WM, WN, BIGC, IN16, 4>(a, s);
  return launch_nm<BN, WM, WN, BIGC, IN16, 3>(a, s);
}

// f32 [rows][k] (row stride ld) * scale -> hi/lo f16 planes [rows][kp], zero padded to kp
__global__ void split_f16_kernel(const float* __restrict__ src, int64_t rows, int k, int ld, int kp, float scale,
                                 _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
  const int64_t total = rows * (kp / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rrow = i / (kp / 2);
    const int c = (int)(i - rrow * (kp / 2)) * 2;
    const float a = c < k ? src[rrow * ld + c] * scale : 0.f;
    const float b = c + 1 < k ? src[rrow * ld + c + 1] * scale : 0.f;
    h16x2 h, l;
    vfml_split2(a, b, h, l);
    *reinterpret_cast<h16x2*>(hi + rrow * kp + c) = h;
    *reinterpret_cast<h16x2*>(lo + rrow * kp + c) = l;
  }
}

// f32 rows -> split rows, one quad (4 channels) per thread
__global__ void to_s16_kernel(const float* __restrict__ src, int64_t rows, int c, int lds, float* __restrict__ dst,
                              int ldd, float scale) {
  const int q4 = c / 4;
  const int64_t total = rows * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / q4;
    const int col = (int)(i - row * q4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + row * lds + col) * scale;
    U8 hi, lo;
    split4(v, hi, lo, 0);
    char* u = reinterpret_cast<char*>(dst + row * ldd + (col & ~7)) + (col & 4) * 2;
    *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hi.v, hi.v, 0, 1, 2, 3));
    *reinterpret_cast<uint2*>(u + 16) = __builtin_bit_cast(uint2, __builtin_shufflevector(lo.v, lo.v, 0, 1, 2, 3));
  }
}

// [rows][c] f32 -> planes [c][kp] of the transpose: 64x64 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src, int rows, int c, int ld,
                                                              int kp, float scale, _Float16* __restrict__ hi,
                                                              _Float16* __restrict__ lo) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const int r = r0 + rr, col = c0 + cc;
    tile[rr][cc] = (r < rows && col < c) ? src[(int64_t)r * ld + col] * scale : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int cc = i >> 5, kk = (i & 31) * 2;
    const int col = c0 + cc, k = r0 + kk;
    if (col >= c || k >= kp) continue;
    const float a = tile[kk][cc], b = tile[kk + 1][cc];
    h16x2 h, l;
    vfml_split2(a, b, h, l);
    *reinterpret_cast<h16x2*>(hi + (int64_t)col * kp + k) = h;
    *reinterpret_cast<h16x2*>(lo + (int64_t)col * kp + k) = l;
  }
}

// one workgroup per row; three sweeps (max, sum of exp, write) over a row that stays in L2
__global__ __launch_bounds__(256) void softmax_rows_s16_kernel(const float* __restrict__ x, int cols, int64_t ld_in,
                                                               float* __restrict__ out, int64_t ld_out, float scale) {
  __shared__ float red[4];
  const float* row = x + (int64_t)blockIdx.x * ld_in;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  float m = -INFINITY;
  for (int c = t; c < cols; c += 256) m = fmaxf(m, row[c]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = t; c < cols; c += 256) s += expf(row[c] - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  {
    // (the same sum in the same order, as two plain v_add_f32: left to the vectoriser this becomes a v_pk_add_f32 that is the
    // first reader of a ds_read2_b32 pair - the sequence that read a stale register in the lookup kernel when another
    // kernel's MFMAs shared the SIMD, profiles/r02_kernel_anatomy.md section 7)
    const float r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
    float s01, s23;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s01) : "v"(r0), "v"(r1));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s23) : "v"(r2), "v"(r3));
    s = s01 + s23;
  }
  const float inv = scale / s;
  float* orow = out + (int64_t)blockIdx.x * ld_out;
  const int nq = (int)(ld_out / 4);
  for (int q4 = t; q4 < nq; q4 += 256) {
    const int c = q4 * 4;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < cols ? expf(row[c + e] - m) * inv : 0.f;
    U8 hh, ll;
    split4(v, hh, ll, 0);
    char* u = reinterpret_cast<char*>(orow + (c & ~7)) + (c & 4) * 2;
    *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hh.v, hh.v, 0, 1, 2, 3));
    *reinterpret_cast<uint2*>(u + 16) = __builtin_bit_cast(uint2, __builtin_shufflevector(ll.v, ll.v, 0, 1, 2, 3));
  }
}

// The same for rows of at most 256 * 4 * SOFTMAX_REG_QUADS columns: the row is read once (16-byte loads, all in
// flight together) and stays in registers for the max, the sum of exp and the write - one exp per element instead
// of two, no second and third sweep through L2.
// PLAIN16: the result leaves as one round-to-nearest f16 per element (rows of ld_out halves) instead of split rows.
constexpr int SOFTMAX_REG_QUADS = 32;
template <bool PLAIN16>
__global__ __launch_bounds__(256) void softmax_rows_s16_reg_kernel(const float* __restrict__ x, int cols, int64_t ld_in,
                                                                   float* __restrict__ out, int64_t ld_out, float scale) {
  __shared__ float red[4];
  const float* row = x + (int64_t)blockIdx.x * ld_in;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const bool vec = (ld_in % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
  f32x4 v[SOFTMAX_REG_QUADS];
#pragma unroll
  for (int i = 0; i < SOFTMAX_REG_QUADS; ++i) {
    const int c = (t + 256 * i) * 4;
    if (vec && c + 3 < cols) {
      v[i] = *reinterpret_cast<const f32x4*>(row + c);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = c + e < cols ? row[c + e] : -INFINITY;
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < SOFTMAX_REG_QUADS; ++i) m = fmaxf(fmaxf(fmaxf(m, v[i][0]), fmaxf(v[i][1], v[i][2])), v[i][3]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SOFTMAX_REG_QUADS; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = expf(v[i][e] - m);          // exp(-inf) = 0 past the row's end
      s += v[i][e];
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  {
    // (the same sum in the same order, as two plain v_add_f32: left to the vectoriser this becomes a v_pk_add_f32 that is the
    // first reader of a ds_read2_b32 pair - the sequence that read a stale register in the lookup kernel when another
    // kernel's MFMAs shared the SIMD, profiles/r02_kernel_anatomy.md section 7)
    const float r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
    float s01, s23;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s01) : "v"(r0), "v"(r1));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s23) : "v"(r2), "v"(r3));
    s = s01 + s23;
  }
  const float inv = scale / s;
  float* orow = out + (int64_t)blockIdx.x * ld_out;
#pragma unroll
  for (int i = 0; i < SOFTMAX_REG_QUADS; ++i) {
    const int c = (t + 256 * i) * 4;
    if constexpr (PLAIN16) {
      if (c < ld_out) {
        const f32x4 pv = v[i] * inv;
        typedef _Float16 h16x4_ __attribute__((ext_vector_type(4)));
        const h16x4_ h = {(_Float16)pv[0], (_Float16)pv[1], (_Float16)pv[2], (_Float16)pv[3]};
        *reinterpret_cast<h16x4_*>(reinterpret_cast<_Float16*>(out) + (int64_t)blockIdx.x * ld_out + c) = h;
      }
    } else if (c < ld_out) {
      U8 hh, ll;
      split4(v[i] * inv, hh, ll, 0);
      char* u = reinterpret_cast<char*>(orow + (c & ~7)) + (c & 4) * 2;
      *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hh.v, hh.v, 0, 1, 2, 3));
      *reinterpret_cast<uint2*>(u + 16) = __builtin_bit_cast(uint2, __builtin_shufflevector(ll.v, ll.v, 0, 1, 2, 3));
    }
  }
}

// [rows][c] f32 -> split ROWS of the transpose times scale: dst row = source column, its channels = source rows
// (64 x 64 tiles through LDS, both sides coalesced): the activation operand of  out^T = V^T . A^T
__global__ __launch_bounds__(256) void transpose_to_s16_kernel(const float* __restrict__ src, int rows, int c, int ld,
                                                               float scale, _Float16* __restrict__ dst, int64_t ld_dst_h) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const int r = r0 + rr, col = c0 + cc;
    tile[rr][cc] = (r < rows && col < c) ? src[(int64_t)r * ld + col] * scale : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int cc = i >> 5, kk = (i & 31) * 2;
    const int col = c0 + cc, k = r0 + kk;
    if (col >= c || 2 * (int64_t)k >= ld_dst_h) continue;       // (rows past `rows` inside the row stride: zeros)
    const float a = tile[kk][cc], b = tile[kk + 1][cc];
    h16x2 h, l;
    vfml_split2(a, b, h, l);
    _Float16* u = dst + (int64_t)col * ld_dst_h + (k >> 3) * 16 + (k & 7);   // unit k/8: 8 hi halves, then 8 lo halves
    *reinterpret_cast<h16x2*>(u) = h;
    *reinterpret_cast<h16x2*>(u + 8) = l;
  }
}

// out (split rows) = aux (split rows) + scale * x (f32), c channels per row
__global__ void add_to_s16_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ aux, int64_t ld_aux,
                                  float* __restrict__ out, int64_t ld_out, int64_t rows, int c, float scale) {
  const int q4 = c / 4;
  const int64_t total = rows * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / q4;
    const int col = (int)(i - row * q4) * 4;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * ldx + col);
    const char* ua = reinterpret_cast<const char*>(aux + row * ld_aux + (col & ~7)) + (col & 4) * 2;
    const h16x2 h0 = *reinterpret_cast<const h16x2*>(ua), h1 = *reinterpret_cast<const h16x2*>(ua + 4);
    const h16x2 l0 = *reinterpret_cast<const h16x2*>(ua + 16), l1 = *reinterpret_cast<const h16x2*>(ua + 20);
    f32x4 v;
    v[0] = ((float)h0[0] + (float)l0[0]) + scale * xv[0];
    v[1] = ((float)h0[1] + (float)l0[1]) + scale * xv[1];
    v[2] = ((float)h1[0] + (float)l1[0]) + scale * xv[2];
    v[3] = ((float)h1[1] + (float)l1[1]) + scale * xv[3];
    U8 hi, lo;
    split4(v, hi, lo, 0);
    char* u = reinterpret_cast<char*>(out + row * ld_out + (col & ~7)) + (col & 4) * 2;
    *reinterpret_cast<uint2*>(u) = __builtin_bit_cast(uint2, __builtin_shufflevector(hi.v, hi.v, 0, 1, 2, 3));
    *reinterpret_cast<uint2*>(u + 16) = __builtin_bit_cast(uint2, __builtin_shufflevector(lo.v, lo.v, 0, 1, 2, 3));
  }
}

}  // namespace

extern "C" int vfml_transpose_split_f16(const float* src, int rows, int c, int ld, float scale, void* hi, void* lo,
                                        int kp, void* stream) {
  VFML_REQUIRE(src && hi && lo && rows > 0 && c > 0 && ld >= c, "vfml_transpose_split_f16: bad argument");
  VFML_REQUIRE(kp >= rows && kp % 32 == 0 && scale > 0.f, "vfml_transpose_split_f16: kp must be rows rounded up to 32");
  VFML_REQUIRE(vfml_aligned16(hi) && vfml_aligned16(lo), "vfml_transpose_split_f16: alignment");
  hipLaunchKernelGGL(transpose_split_kernel, dim3((kp + 63) / 64, (c + 63) / 64), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, rows, c, ld, kp, scale, (_Float16*)hi, (_Float16*)lo);
  return vfml_check_launch("vfml_transpose_split_f16");
}

extern "C" int vfml_softmax_rows_s16(const float* x, int64_t rows, int cols, int64_t ld_in, float* out, int64_t ld_out,
                                     float scale, void* stream) {
  VFML_REQUIRE(scale >= 1.0f && scale <= 32768.0f, "vfml_softmax_rows_s16: scale %g out of [1, 2^15]", (double)scale);
  VFML_REQUIRE(x && out && rows > 0 && rows < (1ll << 31) && cols > 0 && ld_in >= cols && ld_out >= cols && ld_out % 8 == 0,
               "vfml_softmax_rows_s16: bad shape (ld_out %% 8 == 0, ld_out >= cols)");
  VFML_REQUIRE((reinterpret_cast<uintptr_t>(out) & 31u) == 0, "vfml_softmax_rows_s16: out must be 32-byte aligned");
  static const int sweep = getenv("VFML_SOFTMAX_SWEEPS") ? atoi(getenv("VFML_SOFTMAX_SWEEPS")) : 0;
  if (ld_out <= 256 * 4 * SOFTMAX_REG_QUADS && !sweep)
    hipLaunchKernelGGL(softmax_rows_s16_reg_kernel<false>, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, cols, ld_in, out, ld_out, scale);
  else
    hipLaunchKernelGGL(softmax_rows_s16_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                       cols, ld_in, out, ld_out, scale);
  return vfml_check_launch("vfml_softmax_rows_s16");
}

extern "C" int vfml_softmax_rows_f16(const float* x, int64_t rows, int cols, int64_t ld_in, void* out, int64_t ld_out,
                                     float scale, void* stream) {
  VFML_REQUIRE(x && out && rows > 0 && rows < (1ll << 31) && cols > 0 && ld_in >= cols && ld_out >= cols && ld_out % 8 == 0,
               "vfml_softmax_rows_f16: bad shape (ld_out %% 8 == 0, ld_out >= cols)");
  VFML_REQUIRE(ld_out <= 256 * 4 * SOFTMAX_REG_QUADS, "vfml_softmax_rows_f16: rows of at most %d columns", 256 * 4 * SOFTMAX_REG_QUADS);
  VFML_REQUIRE(vfml_aligned16(out), "vfml_softmax_rows_f16: out must be 16-byte aligned");
  VFML_REQUIRE(scale >= 1.0f && scale <= 32768.0f, "vfml_softmax_rows_f16: scale %g out of [1, 2^15]", (double)scale);
  hipLaunchKernelGGL(softmax_rows_s16_reg_kernel<true>, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, cols, ld_in, (float*)out, ld_out, scale);
  return vfml_check_launch("vfml_softmax_rows_f16");
}

extern "C" int vfml_transpose_to_s16(const float* src, int rows, int c, int ld, float scale, float* dst, int64_t ld_dst,
                                     void* stream) {
  VFML_REQUIRE(src && dst && rows > 0 && c > 0 && ld >= c && scale > 0.f, "vfml_transpose_to_s16: bad argument");
  VFML_REQUIRE(ld_dst % 32 == 0 && ld_dst >= rows && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0,
               "vfml_transpose_to_s16: ld_dst must be rows rounded up to 32, dst 32-byte aligned");
  // every 64-row block that touches the row stride is swept, so the pad channels are written (zeros)
  hipLaunchKernelGGL(transpose_to_s16_kernel, dim3((unsigned)((ld_dst + 63) / 64), (c + 63) / 64), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, rows, c, ld, scale, (_Float16*)dst, 2 * ld_dst);
  return vfml_check_launch("vfml_transpose_to_s16");
}

extern "C" int vfml_add_to_s16(const float* x, int64_t ldx, const float* aux, int64_t ld_aux, float* out, int64_t ld_out,
                               int64_t rows, int c, float scale, void* stream) {
  VFML_REQUIRE(x && aux && out && rows > 0 && c > 0 && c % 8 == 0, "vfml_add_to_s16: bad argument (c %% 8 == 0)");
  VFML_REQUIRE(ldx % 4 == 0 && ldx >= c && ld_aux % 8 == 0 && ld_out % 8 == 0 && vfml_aligned16(x) &&
               (reinterpret_cast<uintptr_t>(aux) & 31u) == 0 && (reinterpret_cast<uintptr_t>(out) & 31u) == 0,
               "vfml_add_to_s16: alignment (x 16 bytes / ldx %% 4, split rows 32 bytes / ld %% 8)");
  const int64_t total = rows * (c / 4);
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(add_to_s16_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, aux,
                     ld_aux, out, ld_out, rows, c, scale);
  return vfml_check_launch("vfml_add_to_s16");
}

extern "C" int vfml_to_s16(const float* src, int64_t rows, int c, int ld_src, float* dst, int ld_dst, float scale,
                           void* stream) {
  VFML_REQUIRE(src && dst && rows > 0 && c > 0 && c % 4 == 0 && ld_src >= c && ld_src % 4 == 0 && ld_dst % 8 == 0 &&
               ld_dst >= ((c + 7) & ~7), "vfml_to_s16: bad shape (c %% 4, ld_src %% 4, ld_dst %% 8)");
  VFML_REQUIRE(vfml_aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0, "vfml_to_s16: alignment");
  const int64_t total = rows * (c / 4);
  int64_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(to_s16_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, rows, c,
                     ld_src, dst, ld_dst, scale);
  return vfml_check_launch("vfml_to_s16");
}

extern "C" int vfml_split_f16(const float* src, int64_t rows, int k, int ld, float scale, void* hi, void* lo, int kp,
                              void* stream) {
  VFML_REQUIRE(src && hi && lo, "vfml_split_f16: null pointer");
  VFML_REQUIRE(rows > 0 && k > 0 && ld >= k && kp >= k && kp % 32 == 0, "vfml_split_f16: bad rows/k/ld/kp (kp%%32==0)");
  VFML_REQUIRE(scale > 0.f, "vfml_split_f16: scale must be positive (a power of two keeps the split exact)");
  VFML_REQUIRE(vfml_aligned16(hi) && vfml_aligned16(lo), "vfml_split_f16: hi/lo must be 16-byte aligned");
  const int64_t total = rows * (kp / 2);
  int64_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(split_f16_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, rows, k,
                     ld, kp, scale, (_Float16*)hi, (_Float16*)lo);
  return vfml_check_launch("vfml_split_f16");
}

// out[r][0 .. 4 q4) += add[r][0 .. 4 q4): the second half of a split K axis (vfml_conv_desc.ksplit_ws)
__global__ void add_rows_kernel(float* __restrict__ out, const float* __restrict__ add, int rows, int q4, int ld) {
  const int64_t total = (int64_t)rows * q4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q4;
    const int64_t o = r * ld + (i - r * q4) * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(out + o), y = *reinterpret_cast<const f32x4*>(add + o);
    *reinterpret_cast<f32x4*>(out + o) = x + y;
  }
}

extern "C" int vfml_conv2d_split(const vfml_conv_desc* d, const void* w_hi, const void* w_lo, int kp, float w_scale,
                                 int in_fmt, int out_fmt, int aux_fmt, int k_order, void* stream) {
  VFML_REQUIRE(d != nullptr, "vfml_conv2d_split: null descriptor");
  {
    const int pbits = d->flags & (VFML_CONV_MFMA2 | VFML_CONV_MFMA1 | VFML_CONV_MFMA2A);
    VFML_REQUIRE((d->flags & ~(VFML_CONV_SWAP_CROSS | VFML_CONV_MFMA2 | VFML_CONV_MFMA1 | VFML_CONV_MFMA2A | VFML_CONV_PER_TAP)) == 0,
                 "vfml_conv2d_split: unknown flag bits");
    VFML_REQUIRE((pbits & (pbits - 1)) == 0 && !((d->flags & VFML_CONV_SWAP_CROSS) && pbits),
                 "vfml_conv2d_split: VFML_CONV_MFMA2 / _MFMA2A / _MFMA1 / _SWAP_CROSS exclude one another");
  }
  VFML_REQUIRE(in_fmt == VFML_FMT_S16 || ((d->flags & VFML_CONV_SWAP_CROSS) == 0 && d->out_t == nullptr),
               "vfml_conv2d_split: out_t / VFML_CONV_SWAP_CROSS need split-row sources");
  if (d->stats_part) {
    const int64_t hw_out = (int64_t)((d->h + 2 * d->pad_h - d->kh) / d->stride + 1) * ((d->w + 2 * d->pad_w - d->kw) / d->stride + 1);
    const int rows = in_fmt == VFML_FMT_S16 ? VFML_STATS_ROWS_S16 : VFML_STATS_ROWS_F32;
    VFML_REQUIRE(out_fmt == VFML_FMT_F32 && d->epilogue == VFML_EPI_NONE && !d->addend &&
                 (d->n == 1 || hw_out % rows == 0) && (reinterpret_cast<uintptr_t>(d->stats_part) & 7u) == 0,
                 "vfml_conv2d_split: stats_part needs a plain f32 output, no epilogue / addend, and pixel blocks that do "
                 "not straddle images (n == 1 or output pixels per image %% %d == 0)", rows);
  }
  VFML_REQUIRE(k_order == VFML_KORDER_TAP || ((k_order == VFML_KORDER_CBLOCK || k_order == VFML_KORDER_CBLOCK64) && in_fmt == VFML_FMT_S16),
               "vfml_conv2d_split: bad k_order (channel-block orders need split-row sources)");
  VFML_REQUIRE(k_order != VFML_KORDER_CBLOCK64 || ((d->flags & VFML_CONV_MFMA1) && d->c0 % 64 == 0 && (d->c0 + d->c1) % 64 == 0 && d->cout > 32),
               "vfml_conv2d_split: VFML_KORDER_CBLOCK64 is the weight order of VFML_CONV_MFMA1 calls over whole 64-channel blocks "
               "with more than 32 output channels");
  VFML_REQUIRE((in_fmt == VFML_FMT_F32 || in_fmt == VFML_FMT_S16) &&
               (out_fmt == VFML_FMT_F32 || out_fmt == VFML_FMT_S16 || (out_fmt == VFML_FMT_F16 && in_fmt == VFML_FMT_S16)) &&
               (aux_fmt == VFML_FMT_F32 || aux_fmt == VFML_FMT_S16), "vfml_conv2d_split: bad format selector");
  const bool in16 = in_fmt == VFML_FMT_S16;
  if (in16)
    VFML_REQUIRE(d->c0 % 8 == 0 && d->ld0 % 8 == 0 && d->c1 % 8 == 0 && d->ld1 % 8 == 0 && d->c0 + d->c1 >= 32,
                 "vfml_conv2d_split: split-row sources need channel counts / strides that are multiples of 8 and >= 32 channels");
  if (out_fmt == VFML_FMT_S16)
    VFML_REQUIRE(d->cout % 4 == 0 && d->ldo % 8 == 0 && vfml_aligned16(d->out) && (reinterpret_cast<uintptr_t>(d->out) & 31u) == 0,
                 "vfml_conv2d_split: split-row output needs cout %% 4 == 0, ldo %% 8 == 0 and a 32-byte aligned out");
  if (aux_fmt == VFML_FMT_S16)
    VFML_REQUIRE(d->cout % 4 == 0 && (!d->aux0 || (d->ld_aux0 % 8 == 0 && (reinterpret_cast<uintptr_t>(d->aux0) & 31u) == 0)) &&
                 (!d->aux1 || (d->ld_aux1 % 8 == 0 && (reinterpret_cast<uintptr_t>(d->aux1) & 31u) == 0)) &&
                 (d->epilogue != VFML_EPI_GRU_ZR || d->split % 8 == 0),
                 "vfml_conv2d_split: split-row aux operands need 32-byte aligned bases, ld %% 8 == 0, cout %% 4 == 0");
  VFML_REQUIRE(d->in0 && w_hi && d->out, "vfml_conv2d_split: null in0/w_hi/out");
  const bool bhi = w_lo == nullptr;    // one plain f16 weight plane (GEMM form only, checked below)
  VFML_REQUIRE(!bhi || in_fmt == VFML_FMT_S16, "vfml_conv2d_split: a weight operand without lo plane needs split-row sources");
  VFML_REQUIRE(d->c0 > 0 && d->c0 % 4 == 0 && d->ld0 % 4 == 0 && d->ld0 >= d->c0,
               "vfml_conv2d_split: c0=%d ld0=%d must be multiples of 4 with ld0>=c0", d->c0, d->ld0);
  const bool two = d->in1 != nullptr;
  VFML_REQUIRE(two ? (d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && d->ld1 >= d->c1) : d->c1 == 0,
               "vfml_conv2d_split: c1=%d ld1=%d inconsistent with in1", d->c1, d->ld1);
  VFML_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "vfml_conv2d_split: empty problem");
  VFML_REQUIRE(d->kh > 0 && d->kw > 0 && d->kh * d->kw < 64 && d->stride > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "vfml_conv2d_split: bad kernel geometry (kh*kw must be < 64)");
  VFML_REQUIRE(d->ldo >= d->cout, "vfml_conv2d_split: ldo=%d < cout=%d", d->ldo, d->cout);
  VFML_REQUIRE(vfml_aligned16(d->in0) && vfml_aligned16(w_hi) && (bhi || vfml_aligned16(w_lo)) && (!two || vfml_aligned16(d->in1)),
               "vfml_conv2d_split: in0/in1/w_hi/w_lo must be 16-byte aligned");
  VFML_REQUIRE(w_scale > 0.f, "vfml_conv2d_split: w_scale must be the positive scale given to vfml_split_f16");
  const int K = d->kh * d->kw * (d->c0 + d->c1);
  if (k_order == VFML_KORDER_CBLOCK)
    VFML_REQUIRE(kp == d->kh * d->kw * ((d->c0 + d->c1 + BK - 1) / BK * BK),
                 "vfml_conv2d_split: kp=%d must be kh*kw*roundup32(c0+c1) in channel-block order", kp);
  else if (k_order == VFML_KORDER_CBLOCK64)
    VFML_REQUIRE(kp == K, "vfml_conv2d_split: kp=%d must be kh*kw*(c0+c1) in 64-channel-block order", kp);
  else
    VFML_REQUIRE(kp >= K && kp % BK == 0 && kp < K + BK, "vfml_conv2d_split: kp=%d must be K=%d rounded up to %d", kp, K, BK);
  const int ho = (d->h + 2 * d->pad_h - d->kh) / d->stride + 1;
  const int wo = (d->w + 2 * d->pad_w - d->kw) / d->stride + 1;
  VFML_REQUIRE(ho > 0 && wo > 0, "vfml_conv2d_split: empty output");
  const int64_t M64 = (int64_t)d->n * ho * wo;
  VFML_REQUIRE(M64 < (1ll << 31) - BM, "vfml_conv2d_split: too many output pixels");
  // 1x1 / stride 1 over ONE split-row source (GEMM rows): the LDS-DMA kernel bases its descriptor at each
  // tile's first row, so the source may be of any size (the MemFlow attention matrix is 4.2 GB)
  const bool tilebase = in16 && !two && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 &&
                        (int64_t)256 * d->ld0 * 4 < (1ll << 31);
  if (!tilebase) {
    const int64_t px = (int64_t)d->n * d->h * d->w + (int64_t)(d->pad_h + 1) * d->w + d->pad_w;
    VFML_REQUIRE(px * d->ld0 * 4 < (1ll << 30) && (!two || px * d->ld1 * 4 < (1ll << 30)),
                 "vfml_conv2d_split: a source spans >= 1 GiB");
  }
  {
    VFML_REQUIRE(bhi ? (int64_t)d->cout * kp * 2 <= 0x7ffffff0ll : ((int64_t)d->cout + 128) * kp * 2 < (1ll << 30),
                 "vfml_conv2d_split: weight planes too large");
  }
  if (d->epilogue == VFML_EPI_GRU_ZR || d->epilogue == VFML_EPI_TANH_RELU)
    VFML_REQUIRE(d->split > 0 && d->split < d->cout && d->split % 4 == 0,
                 "vfml_conv2d_split: split=%d must be a multiple of 4 inside (0, cout)", d->split);
  if (d->epilogue == VFML_EPI_GRU_ZR || d->epilogue == VFML_EPI_ADD_AUX)
    VFML_REQUIRE(d->aux0, "vfml_conv2d_split: this epilogue needs aux0");
  if (d->epilogue == VFML_EPI_GRU_Q) VFML_REQUIRE(d->aux0 && d->aux1, "vfml_conv2d_split: GRU_Q needs aux0 and aux1");
  VFML_REQUIRE(d->epilogue >= VFML_EPI_NONE && d->epilogue <= VFML_EPI_ADD_AUX, "vfml_conv2d_split: bad epilogue");

  SplitArgs a;
  a.wbase = nullptr; a.whi_off = a.wlo_off = a.bytesb = 0; a.korder = k_order; a.direct = 0; a.fastk = 0; a.abias = 0; a.src1_delta = 0; a.out_t = nullptr; a.ld_out_t = 0; a.cswap = 0; a.bhi = 0;
  a.stats_part = d->stats_part;
  a.ksplit = 1; a.out_k1 = nullptr; a.out_t_k1 = nullptr;
  static const int fast_epi = getenv("VFML_FAST_EPI") ? atoi(getenv("VFML_FAST_EPI")) : 1;
  a.fast_epi = fast_epi;
  a.nm = (d->flags & VFML_CONV_MFMA1) ? 1 : (d->flags & VFML_CONV_MFMA2A) ? (bhi ? 1 : 4) : ((d->flags & VFML_CONV_MFMA2) || bhi) ? 2 : 3;
  a.pointwise = d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0;
  // one buffer descriptor serves both sources: they must lie in one allocation (within 1 GiB)
  const float* base = (two && d->in1 < d->in0) ? d->in1 : d->in0;
  const int64_t e0 = (d->in0 - base) + ((int64_t)d->n * d->h * d->w - 1) * d->ld0 + d->c0;
  const int64_t e1 = two ? (d->in1 - base) + ((int64_t)d->n * d->h * d->w - 1) * d->ld1 + d->c1 : 0;
  VFML_REQUIRE(tilebase || (e0 > e1 ? e0 : e1) * 4 < (1ll << 31),
               "vfml_conv2d_split: in0 and in1 must be slices of one buffer (< 2 GiB apart)");
  a.in0 = base; a.in1 = base;
  a.d0off = (int)(d->in0 - base); a.d1off = two ? (int)(d->in1 - base) : 0;
  const bool fits_one = (e0 > e1 ? e0 : e1) * 4 < (1ll << 31) && (int64_t)d->n * d->h * d->w * d->ld0 * 4 < (1ll << 30);
  a.bytes0 = fits_one ? (int)((e0 > e1 ? e0 : e1) * 4) : 0;
  a.tilebase = 0;
  a.proj_w = nullptr; a.proj_lo_off = a.proj_bytes = a.proj_n = a.proj_kp = 0; a.proj_inv = 1.f; a.proj_out = nullptr; a.ld_proj = 0;
  a.whi = (const _Float16*)w_hi; a.wlo = (const _Float16*)w_lo; a.bias = d->bias;
  a.aux0 = d->aux0; a.aux1 = d->aux1; a.out = d->out;
  a.addend = d->addend; a.ld_addend = d->ld_addend;
  a.addend_ind = d->addend_ind;
  VFML_REQUIRE(!d->addend_ind || (d->addend && (reinterpret_cast<uintptr_t>(d->addend_ind) & 7u) == 0),
               "vfml_conv2d_split: addend_ind goes with an addend of the same shape and alignment (what the call validates) and is 8-byte aligned");
  a.c0 = d->c0; a.ld0 = d->ld0; a.c1 = d->c1; a.ld1 = two ? d->ld1 : d->ld0; a.ctot = d->c0 + d->c1;
  a.H = d->h; a.W = d->w; a.ho = ho; a.wo = wo;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.M = (int)M64; a.K = K; a.Kp = kp; a.cout = d->cout;
  a.bytesw = (int)((int64_t)d->cout * kp * 2);
  a.ldo = d->ldo; a.ld_aux0 = d->ld_aux0; a.ld_aux1 = d->ld_aux1;
  a.epilogue = d->epilogue; a.split = (d->epilogue == VFML_EPI_GRU_ZR || d->epilogue == VFML_EPI_TANH_RELU) ? d->split : 0;
  a.out_scale = d->out_scale; a.w_inv = 1.0f / w_scale;
  a.mtiles = (a.M + BM - 1) / BM;
  a.vec_ok = vfml_aligned16(d->out) && d->ldo % 4 == 0 &&
             (!d->aux0 || (vfml_aligned16(d->aux0) && d->ld_aux0 % 4 == 0)) &&
             (!d->aux1 || (vfml_aligned16(d->aux1) && d->ld_aux1 % 4 == 0)) &&
             (!d->addend || (vfml_aligned16(d->addend) && d->ld_addend % 4 == 0));
  VFML_REQUIRE(!d->addend || d->ld_addend >= d->cout, "vfml_conv2d_split: ld_addend=%d < cout", d->ld_addend);
  a.out16 = out_fmt == VFML_FMT_S16;
  a.out_h16 = out_fmt == VFML_FMT_F16;
  a.aux16 = aux_fmt == VFML_FMT_S16;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool bigc = a.ctot >= BK;
  // Tile width: 128 columns per workgroup unless 64-wide tiles use the machine better.  Efficiency
  // model = (useful columns / padded columns) x (workgroups / slots of the last partial round) x a
  // 0.7 handicap for the narrower tile (half the MFMAs per loaded A element; measured: 64-wide tiles lose more than the tail round gains on the 1080p shapes); 2 workgroups per CU.
  int bn = d->cout > 64 ? 128 : (d->cout > 32 ? 64 : 32);
  if (d->cout > 64) {
    auto eff = [&](int w) {
      const int nt = (d->cout + w - 1) / w;
      const int64_t wg = (int64_t)a.mtiles * nt, slots = 512;
      const int64_t rounds = (wg + slots - 1) / slots;
      return ((double)d->cout / (nt * w)) * ((double)wg / (rounds * slots)) * (w == 64 ? 0.7 : 1.0);
    };
    static const int force = getenv("VFML_BN") ? atoi(getenv("VFML_BN")) : 0;
    if (force == 64 || (force == 0 && eff(64) > eff(128))) bn = 64;
  }
  if (in16) {   // split-row sources: every slice is a multiple of 8 channels and >= one K step wide
    // LDS-DMA kernel when both weight planes fit one descriptor window (< 1 GiB)
    const char* ph = (const char*)w_hi;
    const char* pl = bhi ? ph : (const char*)w_lo;
    const char* wb = ph < pl ? ph : pl;
    const int64_t ext = (ph < pl ? pl - ph : ph - pl) + (int64_t)d->cout * kp * 2;
    // (a single plane may span up to 2 GiB: its lanes' out-of-range marker is 0x7ffffff0 instead of 1 GiB)
    const bool dma_ok = bhi ? ext <= 0x7ffffff0ll : ext < (1ll << 30);
    if (dma_ok) {   // every split-row source goes through the LDS-DMA kernel
      a.wbase = wb; a.whi_off = (int)(ph - wb); a.wlo_off = (int)(pl - wb); a.bytesb = (int)ext;
      a.bhi = bhi ? 1 : 0;
      a.tilebase = tilebase;
      {
        static const int no_fastk = getenv("VFML_NO_FASTK") ? atoi(getenv("VFML_NO_FASTK")) : 0;
        const int64_t abias = ((int64_t)d->pad_h * d->w + d->pad_w) * d->ld0 * 4;
        // (for a 1x1 convolution over whole 32-channel blocks the two K orders are the same bytes)
        const bool cblock = k_order == VFML_KORDER_CBLOCK || k_order == VFML_KORDER_CBLOCK64 ||
                            (a.pointwise && (d->c0 + d->c1) % BK == 0);
        a.fastk = !no_fastk && cblock && d->c0 % BK == 0 && (d->c0 + d->c1) % BK == 0 &&
                  (!two || (d->ld1 == d->ld0 && a.d1off >= a.d0off)) && d->kh * d->kw <= 32 &&
                  (int64_t)a.bytes0 + abias < (1ll << 31);
        VFML_REQUIRE(k_order != VFML_KORDER_CBLOCK64 || a.fastk,
                     "vfml_conv2d_split: VFML_KORDER_CBLOCK64 needs the uniform-step loader (one row stride for both sources, kh*kw <= 32)");
        if (a.fastk) a.korder = VFML_KORDER_CBLOCK;
        // one MFMA per product over whole 64-channel blocks: 64-channel steps of hi halves (NM 5) - for 1x1 kernels in
        // any weight order (the K axis is the channel axis), else with the weights in 64-channel-block order
        static const int no_h64 = getenv("VFML_NO_H64") ? atoi(getenv("VFML_NO_H64")) : 0;
        // (cout > 32: the 128 x 32 tile of narrower outputs has no uniform-step instantiation)
        if (a.nm == 1 && a.fastk && !no_h64 && d->c0 % 64 == 0 && (d->c0 + d->c1) % 64 == 0 && d->cout > 32 &&
            (k_order == VFML_KORDER_CBLOCK64 || a.pointwise))
          a.nm = 5;
        VFML_REQUIRE(k_order != VFML_KORDER_CBLOCK64 || a.nm == 5, "vfml_conv2d_split: VFML_KORDER_CBLOCK64 weights need the 64-channel-step kernel (VFML_NO_H64 is set?)");
        a.abias = a.fastk ? (int)abias : 0;
        a.src1_delta = two ? (a.d1off - a.d0off) * 4 : 0;
      }
      static const int direct_min = getenv("VFML_DIRECT_MIN") ? atoi(getenv("VFML_DIRECT_MIN")) : 1024;
      // (VFML_FMT_F16 outputs exist in this form only: any width)
      a.direct = (d->epilogue == VFML_EPI_NONE || d->epilogue == VFML_EPI_RELU) && !d->addend &&
                 (out_fmt == VFML_FMT_F32 || out_fmt == VFML_FMT_F16) && !d->stats_part &&
                 (d->cout >= direct_min || out_fmt == VFML_FMT_F16) && d->cout % 4 == 0 && d->ldo % 4 == 0 && vfml_aligned16(d->out) &&
                 (!d->bias || vfml_aligned16(d->bias));
      VFML_REQUIRE(out_fmt != VFML_FMT_F16 || (a.direct && a.fastk),
                   "vfml_conv2d_split: VFML_FMT_F16 outputs are written by the GEMM form only (1x1 over whole 32-channel blocks, no "
                   "addend / activation beyond ReLU, cout %% 4 == 0, ldo %% 4 == 0, 16-byte aligned out)");
      if (d->flags & VFML_CONV_SWAP_CROSS) {
        VFML_REQUIRE(a.direct && a.fastk, "vfml_conv2d_split: VFML_CONV_SWAP_CROSS is implemented by the GEMM form only "
                                          "(1x1 over whole 32-channel blocks, plain f32 out, cout >= 1024, cout %% 4 == 0)");
        a.cswap = 1;
      }
      if (bhi)
        VFML_REQUIRE(a.direct && a.fastk && !a.cswap, "vfml_conv2d_split: a weight operand without lo plane is implemented by the "
                                                        "GEMM form only (1x1 over whole 32-channel blocks, plain f32 out, cout >= 1024)");
      if (d->out_t) {
        VFML_REQUIRE(a.direct && a.fastk && a.pointwise && !d->bias && d->epilogue == VFML_EPI_NONE && a.M % 4 == 0 &&
                     d->ld_out_t % 4 == 0 && d->ld_out_t >= a.M && vfml_aligned16(d->out_t),
                     "vfml_conv2d_split: out_t needs the GEMM form (1x1 over whole 32-channel blocks, plain f32 out, cout >= 1024, cout %% 4 == 0), no bias / "
                     "epilogue, pixels %% 4 == 0, ld_out_t %% 4 == 0 and >= pixels, 16-byte alignment");
        a.out_t = d->out_t; a.ld_out_t = d->ld_out_t;
      }
      // A GEMM with few tiles and a long K axis (the MemFlow read-out: 254 tiles of 128 x 128 on 512 resident slots, one
      // workgroup per CU streaming 8 MB of its operand): two work items per tile, one per half of K, when the caller gave
      // a workspace for the second half's sums
      bool ksplit = false;
      if (d->ksplit_ws) {
        VFML_REQUIRE(vfml_aligned16(d->ksplit_ws), "vfml_conv2d_split: ksplit_ws must be 16-byte aligned");
        static const int no_ksplit = getenv("VFML_NO_KSPLIT") ? atoi(getenv("VFML_NO_KSPLIT")) : 0;
        const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((d->cout + 127) / 128);
        ksplit = !no_ksplit && a.direct && a.fastk && a.pointwise && out_fmt == VFML_FMT_F32 && !a.cswap && tiles <= 256 &&
                 kp >= 4096;
        if (ksplit) {     // (the workspace holds the primary output's shape, then - with out_t - the transposed one's)
          a.ksplit = 2;
          a.out_k1 = d->ksplit_ws;
          a.out_t_k1 = d->ksplit_ws + (int64_t)a.M * d->ldo;
        }
      }
      if (d->proj_out) {
        // projection epilogue: what the kernel's slab routine is built for
        VFML_REQUIRE(d->epilogue == VFML_EPI_RELU && !d->addend && !d->stats_part && !d->out_t && !a.direct,
                     "vfml_conv2d_split: proj_out goes with VFML_EPI_RELU, no addend / stats_part / out_t");
        VFML_REQUIRE(a.fastk && (a.nm == 3 || a.nm == 4) && d->cout % 128 == 0,
                     "vfml_conv2d_split: proj_out needs the uniform-step loader (channel-block weight order, whole 32-channel "
                     "blocks), the full split product or VFML_CONV_MFMA2A (which then holds for the projection too) and "
                     "cout %% 128 == 0");
        VFML_REQUIRE(d->proj_hi && d->proj_lo && d->proj_n > 0 && d->proj_n <= 48 && d->proj_n % 4 == 0 && d->proj_kp >= d->cout &&
                         d->proj_kp % 8 == 0 && d->ld_proj >= d->proj_n && d->ld_proj % 4 == 0 && vfml_aligned16(d->proj_out) &&
                         vfml_aligned16(d->proj_hi) && vfml_aligned16(d->proj_lo) && d->proj_scale > 0.f,
                     "vfml_conv2d_split: proj_hi / proj_lo [proj_n <= 48, %% 4 == 0][proj_kp >= cout] f16 planes, ld_proj >= proj_n, "
                     "16-byte alignment");
        const char* ph2 = (const char*)d->proj_hi;
        const char* pl2 = (const char*)d->proj_lo;
        const int64_t plane = (int64_t)d->proj_n * d->proj_kp * 2;
        VFML_REQUIRE(pl2 >= ph2 && (pl2 - ph2) + plane < (1ll << 30), "vfml_conv2d_split: proj_lo must follow proj_hi within 1 GiB");
        a.proj_w = ph2; a.proj_lo_off = (int)(pl2 - ph2); a.proj_bytes = (int)((pl2 - ph2) + plane);
        a.proj_n = d->proj_n; a.proj_kp = d->proj_kp; a.proj_inv = 1.0f / d->proj_scale;
        a.proj_out = d->proj_out; a.ld_proj = d->ld_proj;
      }
      const char* tile_env = getenv("VFML_DMA_TILE");   // experiments / tests: "TM,TN,WM,WN" (read per call)
      int cfg = d->cout > 32 ? 2122 : 1141;
      if (d->cout > 64) {
        // 192 x 128, 128 x 192, 128 x 128 or 128 x 64 tiles (two workgroups per CU each).  Cost model:
        // (rounds over the 512 resident slots; a problem that does not fill them is one round, a longer one
        // costs its fractional number of rounds because workgroups of the last round run less contended)
        // x (MFMAs per tile / measured relative efficiency of the tile shape: fewer operand bytes per MFMA
        // on the larger tiles).
        auto cost = [&](int tbm, int tbn, double mf, double eff) {
          const double tiles = (double)((a.M + tbm - 1) / tbm) * (double)((d->cout + tbn - 1) / tbn);
          return (tiles > 512.0 ? tiles / 512.0 : 1.0) * mf / eff;
        };
        const double c3222 = cost(192, 128, 6.0, 1.0), c2322 = cost(128, 192, 6.0, 1.0), c2222 = cost(128, 128, 4.0, 0.93),
                     c2122 = cost(128, 64, 2.0, 0.7);
        cfg = 3222;
        double best = c3222;
        if (c2322 < best) { best = c2322; cfg = 2322; }
        if (c2222 < best) { best = c2222; cfg = 2222; }
        if (c2122 < best) { best = c2122; cfg = 2122; }
        if (a.proj_out && cfg != 3222 && cfg != 2222) cfg = c3222 <= c2222 ? 3222 : 2222;    // (128-column tiles of four waves)
      }
      bool forced = false;
      if (tile_env && d->cout > 32) {
        int tm = 2, tn = 2, wm = 2, wn = 2;
        sscanf(tile_env, "%d,%d,%d,%d", &tm, &tn, &wm, &wn);
        const int want = tm * 1000 + tn * 100 + wm * 10 + wn;
        // (2241 / 2341 exist in the shared-stage kernel only; narrower outputs keep their per-tap shapes otherwise)
        if (d->cout > 64 || want == 2241 || want == 2341) { cfg = want; forced = true; }
        VFML_REQUIRE(!a.proj_out || cfg == 3222 || cfg == 2222, "vfml_conv2d_split: proj_out runs on the 192 x 128 / 128 x 128 tiles (VFML_DMA_TILE)");
      }
      // stride-1 "same" convolutions with a filter row of 2..5 taps: one activation stage per (channel block, tap row),
      // shared by the row's taps (conv_gemm_tapx.hip; VFML_TAPX=0: the per-tap stages of conv_gemm_dma_kernel, for A/B)
      static const int tapx = getenv("VFML_TAPX") ? atoi(getenv("VFML_TAPX")) : 1;
      if (tapx && !(d->flags & VFML_CONV_PER_TAP) && !a.proj_out) {
        // (VFML_TAPX=2: also the three-MFMA calls on the 192 x 128 / 128 x 192 tiles, where the two kernels run level)
        const int tcfg = vfml_detail::tapx_cfg(a, cfg, forced);
        if (tcfg && (tapx >= 2 || forced || a.nm == 5 || tcfg == 2241 || tcfg == 2341)) return vfml_detail::launch_tapx(a, tcfg, s);
      }
      if (a.ksplit == 2) {
        // (the GEMM form is one tile shape; the partial sums of the second half of K are added once the launch is queued)
        const int rc = launch_dma<2, 2, 2, 2>(a, s);
        if (rc) return rc;
        const int64_t quads = (int64_t)a.M * (d->cout / 4);
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((quads + 255) / 256 < 65535 * 16 ? (quads + 255) / 256 : 65535 * 16)), dim3(256), 0, s,
                           d->out, d->ksplit_ws, a.M, d->cout / 4, d->ldo);
        if (d->out_t)      // [cout][ld_out_t] with M valid columns (M % 4 == 0: host check of out_t)
          hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((quads + 255) / 256 < 65535 * 16 ? (quads + 255) / 256 : 65535 * 16)), dim3(256), 0, s,
                             d->out_t, a.out_t_k1, d->cout, a.M / 4, d->ld_out_t);
        return vfml_check_launch("vfml_conv2d_split");
      }
      switch (cfg) {
        case 3222: return launch_dma<3, 2, 2, 2>(a, s);   // 192 x 128, 2 workgroups per CU
        case 2322: return launch_dma<2, 3, 2, 2>(a, s);   // 128 x 192
        case 2122: return launch_dma<2, 1, 2, 2>(a, s);   // 128 x 64
        case 1141: return launch_dma<1, 1, 4, 1>(a, s);   // 128 x 32
        // (the 8-wave shapes of round 1 - 256 x 128, 192 x 256, 256 x 256, one workgroup per CU - measured slower and are
        // no longer built)
        case 2241: case 2341: return launch_dma<2, 1, 2, 2>(a, s);   // (forced shared-stage shapes on a call that kernel does not take)
        default: return launch_dma<2, 2, 2, 2>(a, s);
      }
    }
    vfml_set_error("vfml_conv2d_split: split-row sources need w_hi and w_lo within 1 GiB of each other (one allocation)");
    return 1;
  }
  VFML_REQUIRE(!d->proj_out, "vfml_conv2d_split: proj_out is implemented for split-row (VFML_FMT_S16) sources only");
  if (bn == 128) {
    a.ntiles = (d->cout + 127) / 128;
    return bigc ? launch<128, 2, 2, true, false>(a, s) : launch<128, 2, 2, false, false>(a, s);
  } else if (bn == 64) {
    a.ntiles = (d->cout + 63) / 64;
    return bigc ? launch<64, 2, 2, true, false>(a, s) : launch<64, 2, 2, false, false>(a, s);
  }
  a.ntiles = 1;
  return bigc ? launch<32, 4, 1, true, false>(a, s) : launch<32, 4, 1, false, false>(a, s);
}
