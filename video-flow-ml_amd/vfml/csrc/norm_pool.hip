// HBM-bound helper passes: input normalisation (K1), instance-norm statistics / apply (K2),
// 2x2 average pooling of feature maps (K4 feeder).  All are streaming float4 kernels.
#include <stdlib.h>
#include "vfml_common.h"

namespace {

// ------------------------------------------------------------------ K1: frames -> NHWC4
// One thread per pixel: reads 3 B (u8 HWC) or 3 strided floats (f32 CHW), writes one float4.
__global__ void frames_u8_kernel(const uint8_t* __restrict__ src, int64_t npx, float scale, float shift,
                                 f32x4* __restrict__ dst) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t* s = src + p * 3;
    f32x4 v;
    // x = u8 / 255 in fp32 first (what the host did in the reference), then the affine map.
    v[0] = scale * ((float)s[0] / 255.0f) + shift;
    v[1] = scale * ((float)s[1] / 255.0f) + shift;
    v[2] = scale * ((float)s[2] / 255.0f) + shift;
    v[3] = 0.f;
    dst[p] = v;
  }
}

__global__ void frames_f32_kernel(const float* __restrict__ src, int n, int64_t hw, float scale, float shift,
                                  f32x4* __restrict__ dst) {
  const int64_t npx = (int64_t)n * hw;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = p / hw, q = p - f * hw;
    const float* s = src + f * 3 * hw + q;
    f32x4 v;
    v[0] = scale * s[0] + shift;
    v[1] = scale * s[hw] + shift;
    v[2] = scale * s[2 * hw] + shift;
    v[3] = 0.f;
    dst[p] = v;
  }
}

// ------------------------------------------------------------------ instance-norm statistics
// Pass 1: grid (chunks, n).  A block sweeps its pixel chunk; thread = (pixel lane, float4 channel
// group); sums and sums of squares are carried in double (the CPU reference accumulates in
// double too), reduced across pixel lanes through LDS, one partial per (chunk, channel).
// Pass 2: one thread per (n, channel) folds the chunk partials in fixed order -> {mean, rstd}.
constexpr int STAT_THREADS = 256;

__global__ __launch_bounds__(STAT_THREADS) void instnorm_partial_kernel(const float* __restrict__ x, int hw, int c,
                                                                      int px_per_chunk, double* __restrict__ part) {
  const int cg = c >> 2;                  // float4 groups per pixel
  const int lanes = STAT_THREADS / cg;    // pixel lanes (threads beyond lanes*cg idle)
  const int t = threadIdx.x;
  const int g = t % cg;
  const int pl = t / cg;
  const int n = blockIdx.y;
  const int chunk = blockIdx.x;
  const int p0 = chunk * px_per_chunk;
  const int p1 = min(hw, p0 + px_per_chunk);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (pl < lanes) {
    const f32x4* base = reinterpret_cast<const f32x4*>(x + (int64_t)n * hw * c);
    for (int p = p0 + pl; p < p1; p += lanes) {
      const f32x4 v = base[(int64_t)p * cg + g];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        s[e] += d;
        q[e] += d * d;
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* sh = reinterpret_cast<double*>(smem_raw);  // [lanes][c][2]
  if (pl < lanes) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sh[((int64_t)pl * c + g * 4 + e) * 2 + 0] = s[e];
      sh[((int64_t)pl * c + g * 4 + e) * 2 + 1] = q[e];
    }
  }
  __syncthreads();
  for (int ch = t; ch < c; ch += STAT_THREADS) {
    double ss = 0, qq = 0;
    for (int l = 0; l < lanes; ++l) {
      ss += vfml_lds_f64(&sh[((int64_t)l * c + ch) * 2 + 0]);
      qq += vfml_lds_f64(&sh[((int64_t)l * c + ch) * 2 + 1]);
    }
    double* o = part + (((int64_t)n * gridDim.x + chunk) * c + ch) * 2;
    o[0] = ss;
    o[1] = qq;
  }
}

// One workgroup per (n, channel): threads stride over the chunk partials (four independent loads in flight each),
// then thread 0 folds the FINAL_THREADS sums in thread order (fixed association -> bitwise reproducible).
constexpr int FINAL_THREADS = 256;
__global__ __launch_bounds__(FINAL_THREADS) void instnorm_final_kernel(const double* __restrict__ part, int n, int chunks,
                                                                      int c, int hw, float eps, float* __restrict__ stats) {
  __shared__ double sh[FINAL_THREADS][2];
  const int i = blockIdx.x;   // n * c blocks
  const int nn = i / c, ch = i - nn * c;
  const int t = threadIdx.x;
  const double* base = part + ((int64_t)nn * chunks * c + ch) * 2;
  double s = 0, q = 0;
  int k = t;
  for (; k + 3 * FINAL_THREADS < chunks; k += 4 * FINAL_THREADS) {
    const double* p0 = base + (int64_t)k * c * 2;
    const double* p1 = p0 + (int64_t)FINAL_THREADS * c * 2;
    const double* p2 = p1 + (int64_t)FINAL_THREADS * c * 2;
    const double* p3 = p2 + (int64_t)FINAL_THREADS * c * 2;
    const double a0 = p0[0], b0 = p0[1], a1 = p1[0], b1 = p1[1], a2 = p2[0], b2 = p2[1], a3 = p3[0], b3 = p3[1];
    s += a0; q += b0; s += a1; q += b1; s += a2; q += b2; s += a3; q += b3;
  }
  for (; k < chunks; k += FINAL_THREADS) {
    const double* p = base + (int64_t)k * c * 2;
    s += p[0];
    q += p[1];
  }
  sh[t][0] = s;
  sh[t][1] = q;
  __syncthreads();
  if (t == 0) {
    double ss = 0, qq = 0;
    for (int l = 0; l < FINAL_THREADS; ++l) {
      ss += vfml_lds_f64(&sh[l][0]);
      qq += vfml_lds_f64(&sh[l][1]);
    }
    const double mean = ss / hw;
    double var = qq / hw - mean * mean;
    if (var < 0) var = 0;
    stats[2 * i + 0] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// With many partials per channel (the split-row convolutions leave one per 32 pixels: 16 200 at 540 x 960) the kernel above
// reads 16 bytes per thread at a stride of c * 16 bytes - every 128-byte line eight times, by eight workgroups (23 us per
// call, 30 calls per field).  First pass for that case: one workgroup per (n, 8 channels, slice of the chunks), a thread per
// (channel, lane): eight threads read one 128-byte line, a slice's 32 lane sums are folded in lane order -> [n][slices][c][2]
// partials that the kernel above finishes.  Fixed association: bitwise reproducible.
constexpr int FOLD_SLICES = 64;
__global__ __launch_bounds__(256) void instnorm_fold_kernel(const double* __restrict__ part, int chunks, int c,
                                                            double* __restrict__ out) {
  __shared__ double sh[32][8][2];
  const int cgs = c / 8;
  const int sl = blockIdx.x % FOLD_SLICES, cg = (blockIdx.x / FOLD_SLICES) % cgs, nn = blockIdx.x / (FOLD_SLICES * cgs);
  const int ch = threadIdx.x & 7, lane = threadIdx.x >> 3;
  const int per = (chunks + FOLD_SLICES - 1) / FOLD_SLICES;
  const int k0 = sl * per, k1 = k0 + per < chunks ? k0 + per : chunks;
  const double* base = part + ((int64_t)nn * chunks * c + cg * 8 + ch) * 2;
  double s0 = 0, q0 = 0;
  for (int k = k0 + lane; k < k1; k += 32) {
    const double* p = base + (int64_t)k * c * 2;
    s0 += p[0];
    q0 += p[1];
  }
  sh[lane][ch][0] = s0;
  sh[lane][ch][1] = q0;
  __syncthreads();
  if (lane == 0) {
    double ss = 0, qq = 0;
    for (int l = 0; l < 32; ++l) {
      ss += vfml_lds_f64(&sh[l][ch][0]);
      qq += vfml_lds_f64(&sh[l][ch][1]);
    }
    double* o = out + (((int64_t)nn * FOLD_SLICES + sl) * c + cg * 8 + ch) * 2;
    o[0] = ss;
    o[1] = qq;
  }
}

// ------------------------------------------------------------------ instance-norm apply
// S16: the result - and in MODE 1 the residual, which is an earlier result - are split rows (VFML_FMT_S16): the
// quad g of a pixel sits in unit g / 2, hi halves at byte 8 (g & 1) of the 32-byte unit, lo halves 16 bytes further
template <int MODE, bool S16>  // 0: relu(norm(x)); 1: relu(res + relu(norm(x))); 2: relu(norm(res) + relu(norm(x)))
__global__ void instnorm_apply_kernel(const f32x4* __restrict__ x, const float* __restrict__ stats,
                                      const f32x4* __restrict__ res, const float* __restrict__ rstats, int hw, int c,
                                      int64_t total4, f32x4* __restrict__ out) {
  const int cg = c >> 2;
  const int64_t per_n = (int64_t)hw * cg;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per_n);
    const int g = (int)(i % cg);
    const float* st = stats + ((int64_t)n * c + g * 4) * 2;
    const f32x4 v = x[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf((v[e] - st[2 * e]) * st[2 * e + 1], 0.f);
    const int64_t qoff = (i - g) * 16 + (g >> 1) * 32 + (g & 1) * 8;     // byte offset of the quad's hi halves (S16)
    if (MODE == 1) {
      f32x4 rv;
      if (S16) {
        const char* u = reinterpret_cast<const char*>(res) + qoff;
        const vfml_h16x2 h0 = *reinterpret_cast<const vfml_h16x2*>(u), h1 = *reinterpret_cast<const vfml_h16x2*>(u + 4);
        const vfml_h16x2 l0 = *reinterpret_cast<const vfml_h16x2*>(u + 16), l1 = *reinterpret_cast<const vfml_h16x2*>(u + 20);
        rv[0] = (float)h0[0] + (float)l0[0];
        rv[1] = (float)h0[1] + (float)l0[1];
        rv[2] = (float)h1[0] + (float)l1[0];
        rv[3] = (float)h1[1] + (float)l1[1];
      } else {
        rv = res[i];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(rv[e] + o[e], 0.f);
    } else if (MODE == 2) {
      const f32x4 rv = res[i];
      const float* rs = rstats + ((int64_t)n * c + g * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf((rv[e] - rs[2 * e]) * rs[2 * e + 1] + o[e], 0.f);
    }
    if (S16) {
      vfml_h16x2 h0, h1, l0, l1;
      vfml_split2(o[0], o[1], h0, l0);
      vfml_split2(o[2], o[3], h1, l1);
      char* u = reinterpret_cast<char*>(out) + qoff;
      uint2 hv, lv;
      hv.x = __builtin_bit_cast(unsigned, h0); hv.y = __builtin_bit_cast(unsigned, h1);
      lv.x = __builtin_bit_cast(unsigned, l0); lv.y = __builtin_bit_cast(unsigned, l1);
      *reinterpret_cast<uint2*>(u) = hv;
      *reinterpret_cast<uint2*>(u + 16) = lv;
    } else {
      out[i] = o;
    }
  }
}

// ------------------------------------------------------------------ 2x2 average pool (floor)
__global__ void avgpool2x2_kernel(const f32x4* __restrict__ x, int h, int w, int cg, int ho, int wo, int64_t total4,
                                  f32x4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    int64_t p = i / cg;
    const int ox = (int)(p % wo);
    p /= wo;
    const int oy = (int)(p % ho);
    const int n = (int)(p / ho);
    const f32x4* b = x + (((int64_t)n * h + 2 * oy) * w + 2 * ox) * cg + g;
    const f32x4 a0 = b[0], a1 = b[cg], a2 = b[(int64_t)w * cg], a3 = b[(int64_t)w * cg + cg];
    // same association as at::avg_pool2d's window sum: ((a0 + a1) + a2) + a3, then / 4
    out[i] = (((a0 + a1) + a2) + a3) * 0.25f;
  }
}

inline int grid_for(int64_t items, int block) {
  int64_t g = (items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// Pixel chunks per image: ~1024 px each so that even one 540x960 frame (the steady state of the
// sliding-window cache) spreads over >= 2 workgroups per CU; capped so pass 2 stays short.
int stat_chunks(int hw) {
  int chunks = (hw + 1023) / 1024;
  if (chunks > 1024) chunks = 1024;
  return chunks < 1 ? 1 : chunks;
}

}  // namespace

extern "C" int vfml_frames_to_nhwc4(const void* src, int kind, int n, int H, int W, float scale, float shift,
                                    float* dst, void* stream) {
  VFML_REQUIRE(src && dst, "vfml_frames_to_nhwc4: null pointer");
  VFML_REQUIRE(n > 0 && H > 0 && W > 0, "vfml_frames_to_nhwc4: empty input");
  VFML_REQUIRE(kind == 0 || kind == 1, "vfml_frames_to_nhwc4: kind must be 0 (u8 HWC) or 1 (f32 CHW)");
  VFML_REQUIRE(vfml_aligned16(dst), "vfml_frames_to_nhwc4: dst must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t npx = (int64_t)n * H * W;
  if (kind == 0)
    hipLaunchKernelGGL(frames_u8_kernel, dim3(grid_for(npx, 256)), dim3(256), 0, s, (const uint8_t*)src, npx, scale,
                       shift, (f32x4*)dst);
  else
    hipLaunchKernelGGL(frames_f32_kernel, dim3(grid_for(npx, 256)), dim3(256), 0, s, (const float*)src, n,
                       (int64_t)H * W, scale, shift, (f32x4*)dst);
  return vfml_check_launch("vfml_frames_to_nhwc4");
}

extern "C" int64_t vfml_instnorm_workspace_bytes(int n, int hw, int c) {
  return (int64_t)n * stat_chunks(hw) * c * 2 * sizeof(double);
}

extern "C" int vfml_instnorm_stats(const float* x, int n, int hw, int c, float eps, float* stats, void* workspace,
                                   void* stream) {
  VFML_REQUIRE(x && stats && workspace, "vfml_instnorm_stats: null pointer");
  VFML_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 4 == 0 && c <= 1024, "vfml_instnorm_stats: bad n/hw/c (c%%4==0, c<=1024)");
  VFML_REQUIRE(vfml_aligned16(x), "vfml_instnorm_stats: x must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int chunks = stat_chunks(hw);
  const int px = (hw + chunks - 1) / chunks;
  const int lanes = STAT_THREADS / (c / 4);
  VFML_REQUIRE(lanes >= 1, "vfml_instnorm_stats: too many channels");
  const size_t lds = (size_t)lanes * c * 2 * sizeof(double);
  VFML_REQUIRE(lds <= 64 * 1024, "vfml_instnorm_stats: LDS budget");
  hipLaunchKernelGGL(instnorm_partial_kernel, dim3(chunks, n), dim3(STAT_THREADS), lds, s, x, hw, c, px,
                     (double*)workspace);
  int rc = vfml_check_launch("vfml_instnorm_stats(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(instnorm_final_kernel, dim3(n * c), dim3(FINAL_THREADS), 0, s, (const double*)workspace, n, chunks, c, hw,
                     eps, stats);
  return vfml_check_launch("vfml_instnorm_stats(final)");
}

static bool finalize_folds(int chunks, int c) {
  static const int no_fold = getenv("VFML_NO_NORM_FOLD") ? atoi(getenv("VFML_NO_NORM_FOLD")) : 0;     // (A/B)
  return !no_fold && chunks >= 1024 && c % 8 == 0;
}

extern "C" int64_t vfml_instnorm_finalize_workspace_bytes(int chunks, int c) {
  return finalize_folds(chunks, c) ? (int64_t)FOLD_SLICES * c * 16 : 0;
}

extern "C" int vfml_instnorm_finalize(const double* part, int n, int chunks, int c, int hw, float eps, float* stats,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  VFML_REQUIRE(part && stats && n > 0 && chunks > 0 && c > 0 && hw > 0, "vfml_instnorm_finalize: bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // many partials: fold them slice-wise with coalesced reads first, through the CALLER's workspace.
  // (the route depends on chunks and c only - never on n: a frame's statistics are the same bits whether it is encoded
  // alone or in a batch; batches larger than the workspace go through it a few frames at a time)
  if (finalize_folds(chunks, c)) {
    const int64_t per_frame = (int64_t)FOLD_SLICES * c * 16;
    VFML_REQUIRE(workspace && workspace_bytes >= per_frame && (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0,
                 "vfml_instnorm_finalize: %d partials x %d channels fold through a workspace of >= %lld bytes (16-byte aligned); got %lld",
                 chunks, c, (long long)per_frame, (long long)(workspace ? workspace_bytes : 0));
    const int nmax = (int)(workspace_bytes / per_frame);
    for (int n0 = 0; n0 < n; n0 += nmax) {
      const int nb = n - n0 < nmax ? n - n0 : nmax;
      hipLaunchKernelGGL(instnorm_fold_kernel, dim3(nb * (c / 8) * FOLD_SLICES), dim3(256), 0, st,
                         part + (int64_t)n0 * chunks * c * 2, chunks, c, (double*)workspace);
      hipLaunchKernelGGL(instnorm_final_kernel, dim3(nb * c), dim3(FINAL_THREADS), 0, st, (const double*)workspace, nb,
                         FOLD_SLICES, c, hw, eps, stats + (int64_t)n0 * c * 2);
    }
    return vfml_check_launch("vfml_instnorm_finalize");
  }
  hipLaunchKernelGGL(instnorm_final_kernel, dim3(n * c), dim3(FINAL_THREADS), 0, st, part, n, chunks,
                     c, hw, eps, stats);
  return vfml_check_launch("vfml_instnorm_finalize");
}

extern "C" int vfml_instnorm_apply(const float* x, const float* stats, const float* res, const float* res_stats, int n,
                                   int hw, int c, float* out, int out_fmt, void* stream) {
  VFML_REQUIRE(x && stats && out, "vfml_instnorm_apply: null pointer");
  VFML_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 4 == 0, "vfml_instnorm_apply: bad n/hw/c");
  VFML_REQUIRE(out_fmt == VFML_FMT_F32 || (out_fmt == VFML_FMT_S16 && c % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 31u) == 0 &&
                                            (!res || res_stats || (reinterpret_cast<uintptr_t>(res) & 31u) == 0)),
               "vfml_instnorm_apply: split-row output needs c %% 8 == 0 and 32-byte aligned out (and res)");
  VFML_REQUIRE(!(res_stats && !res), "vfml_instnorm_apply: res_stats without res");
  VFML_REQUIRE(vfml_aligned16(x) && vfml_aligned16(out) && vfml_aligned16(res), "vfml_instnorm_apply: alignment");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int64_t total4 = (int64_t)n * hw * (c / 4);
  const dim3 g(grid_for(total4, 256)), b(256);
  const bool s16 = out_fmt == VFML_FMT_S16;
#define VFML_APPLY(MODE, S16, R, RS)                                                                                      \
  hipLaunchKernelGGL((instnorm_apply_kernel<MODE, S16>), g, b, 0, s, (const f32x4*)x, stats, (const f32x4*)(R), RS, hw, c, \
                     total4, (f32x4*)out)
  if (!res) {
    if (s16) VFML_APPLY(0, true, nullptr, nullptr); else VFML_APPLY(0, false, nullptr, nullptr);
  } else if (!res_stats) {
    if (s16) VFML_APPLY(1, true, res, nullptr); else VFML_APPLY(1, false, res, nullptr);
  } else {
    if (s16) VFML_APPLY(2, true, res, res_stats); else VFML_APPLY(2, false, res, res_stats);
  }
#undef VFML_APPLY
  return vfml_check_launch("vfml_instnorm_apply");
}

extern "C" int vfml_avgpool2x2(const float* x, int n, int h, int w, int c, float* out, void* stream) {
  VFML_REQUIRE(x && out, "vfml_avgpool2x2: null pointer");
  VFML_REQUIRE(n > 0 && h >= 2 && w >= 2 && c > 0 && c % 4 == 0, "vfml_avgpool2x2: bad shape");
  VFML_REQUIRE(vfml_aligned16(x) && vfml_aligned16(out), "vfml_avgpool2x2: alignment");
  const int ho = h / 2, wo = w / 2, cg = c / 4;
  const int64_t total4 = (int64_t)n * ho * wo * cg;
  hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const f32x4*)x, h, w, cg, ho, wo, total4, (f32x4*)out);
  return vfml_check_launch("vfml_avgpool2x2");
}
