// Temporal blend of a frame with its flow-reprojected history (second half): the consumer the
// reference feeds every flow field to in its --taa jobs (effects/taa_processor.py:42-216).  One thread per pixel;
// the op is a gather over the history image: 3 B (frame) + 8 B (flow) + 24 B (history, L2 catches the overlap of the
// four neighbours) read and 24 B written per pixel in the default mode, so HBM-bound at ~60 B per pixel.
//
// Arithmetic follows numpy's type promotion in the reference step by step (comments name the dtype of each
// intermediate), because the history feeds back into itself frame after frame:
//   * coordinates, bilinear weights and the weighted sum are float64 (int64 grid + float32 flow promotes);
//   * the luminance weight is float32 when the history is float32 (the second frame of a sequence) - where it
//     underflows to 0 for luminance differences above ~114 - and float64 afterwards.
#include <type_traits>
#include "vfml_common.h"

namespace {

struct TaaArgs {
  const void* cur; const float* flow; const void* hist; void* out;
  int h, w;
  float alpha32, keep32;     // float32(alpha), float32(1 - alpha)
  double keep64;             // 1 - alpha (Python float)
  double denom64;            // 2 * (sigma^2 * 0.1) + 1e-6
  float denom32;
};

// float32 quotient, correctly rounded: through the float64 quotient (53 >= 2*24 + 2 bits)
__device__ __forceinline__ float div32(float x, float y) { return (float)((double)x / (double)y); }
// float32 exp through the float64 one: correctly rounded (up to double rounding), and it keeps the subnormal results
// numpy returns below exp(-87.3) - they are the whole weight where the luminance jumps by more than ~104
__device__ __forceinline__ float exp32(float x) { return (float)exp((double)x); }
// nan_to_num(nan=0, posinf=hi, neginf=0) then clip(0, hi)
__device__ __forceinline__ double to_range(double v, double hi) {
  if (v != v) return 0.0;
  return fmin(fmax(v, 0.0), hi);
}

inline int blocks_for(int64_t items, int block) {
  int64_t g = (items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

template <typename T>
__device__ __forceinline__ void load3(const T* p, T (&v)[3]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }

// MODE: VFML_TAA_*; HT: history element (float / double); CT: current-frame element (unsigned char / float).
// The output element follows from the two (see vfml_taa_blend).
template <int MODE, typename HT, typename CT>
__global__ void taa_blend_kernel(const TaaArgs a) {
#pragma clang fp contract(off)
  constexpr bool H32 = sizeof(HT) == 4;
  using OT = typename std::conditional<MODE == VFML_TAA_SIMPLE, HT,
                                       typename std::conditional<MODE == VFML_TAA_BILINEAR, float, double>::type>::type;
  const CT* curp = (const CT*)a.cur;
  const HT* hist = (const HT*)a.hist;
  OT* out = (OT*)a.out;
  const int64_t n = (int64_t)a.h * a.w;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const float cur[3] = {(float)curp[3 * p], (float)curp[3 * p + 1], (float)curp[3 * p + 2]};
    if constexpr (MODE == VFML_TAA_SIMPLE) {                 // alpha * cur + (1 - alpha) * history, in the history's type
      HT hv[3];
      load3(hist + 3 * p, hv);
      for (int c = 0; c < 3; ++c) {
        const float head = a.alpha32 * cur[c];               // float32
        if constexpr (H32) out[3 * p + c] = head + a.keep32 * hv[c];
        else out[3 * p + c] = (double)head + a.keep64 * hv[c];
      }
    } else {
      const int y = (int)(p / a.w), x = (int)(p - (int64_t)y * a.w);
      const float2 fl = ((const float2*)a.flow)[p];
      const double px = to_range((double)x + (double)fl.x, (double)(a.w - 1));
      const double py = to_range((double)y + (double)fl.y, (double)(a.h - 1));
      int x0 = (int)floor(px), y0 = (int)floor(py), x1, y1;
      if constexpr (MODE == VFML_TAA_BILATERAL) {            // corners pulled back so that the +1 neighbour exists
        x0 = min(max(x0, 0), a.w - 2);
        y0 = min(max(y0, 0), a.h - 2);
        x1 = x0 + 1;
        y1 = y0 + 1;
      } else {                                               // +1 neighbour clamped
        x1 = min(x0 + 1, a.w - 1);
        y1 = min(y0 + 1, a.h - 1);
      }
      const double wx = px - (double)x0, wy = py - (double)y0;
      HT tap[4][3];
      load3(hist + 3 * ((int64_t)y0 * a.w + x0), tap[0]);
      load3(hist + 3 * ((int64_t)y0 * a.w + x1), tap[1]);
      load3(hist + 3 * ((int64_t)y1 * a.w + x0), tap[2]);
      load3(hist + 3 * ((int64_t)y1 * a.w + x1), tap[3]);
      if constexpr (MODE == VFML_TAA_BILINEAR) {             // float64 sum stored as float32, then a float32 blend
        for (int c = 0; c < 3; ++c) {
          const double v = ((((double)tap[0][c] * (1.0 - wx)) * (1.0 - wy) + ((double)tap[1][c] * wx) * (1.0 - wy)) +
                            ((double)tap[2][c] * (1.0 - wx)) * wy) + ((double)tap[3][c] * wx) * wy;
          out[3 * p + c] = a.alpha32 * cur[c] + a.keep32 * (float)v;
        }
      } else {                                               // spatial weight x luminance similarity to the current pixel
        const float lum = div32((cur[0] + cur[1]) + cur[2], 3.0f);   // np.mean of float32: float32 sum and quotient
        const double sw[4] = {(1.0 - wx) * (1.0 - wy), wx * (1.0 - wy), (1.0 - wx) * wy, wx * wy};
        double wgt[4], total = 0.0;
        for (int k = 0; k < 4; ++k) {
          double cw;
          if constexpr (H32) {
            const float d = lum - div32((tap[k][0] + tap[k][1]) + tap[k][2], 3.0f);
            cw = (double)exp32(div32(-(d * d), a.denom32));
          } else {
            const double d = (double)lum - ((tap[k][0] + tap[k][1]) + tap[k][2]) / 3.0;
            cw = exp(-(d * d) / a.denom64);
          }
          wgt[k] = sw[k] * cw;
          total = k == 0 ? wgt[0] : total + wgt[k];
        }
        if (total == 0.0) total = 1e-6;
        for (int c = 0; c < 3; ++c) {
          const double rep = ((((double)tap[0][c] * wgt[0] + (double)tap[1][c] * wgt[1]) + (double)tap[2][c] * wgt[2]) +
                              (double)tap[3][c] * wgt[3]) / total;
          out[3 * p + c] = (double)(a.alpha32 * cur[c]) + a.keep64 * rep;
        }
      }
    }
  }
}

template <int MODE, typename HT>
void launch_taa(const TaaArgs& a, bool cur_u8, hipStream_t s) {
  const dim3 grid(blocks_for((int64_t)a.h * a.w, 256)), block(256);
  if (cur_u8) hipLaunchKernelGGL((taa_blend_kernel<MODE, HT, unsigned char>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((taa_blend_kernel<MODE, HT, float>), grid, block, 0, s, a);
}

}  // namespace

extern "C" int vfml_taa_blend(const void* current, int cur_type, const float* flow, const void* history, int hist_type,
                              void* out, int out_type, int h, int w, int mode, double alpha, double sigma_color,
                              void* stream) {
  VFML_REQUIRE(current && history && out && h > 0 && w > 0, "vfml_taa_blend: bad argument");
  VFML_REQUIRE(cur_type == VFML_PIX_U8 || cur_type == VFML_PIX_F32, "vfml_taa_blend: current must be u8 or f32");
  VFML_REQUIRE(hist_type == VFML_PIX_F32 || hist_type == VFML_PIX_F64, "vfml_taa_blend: history must be f32 or f64");
  VFML_REQUIRE(mode == VFML_TAA_SIMPLE || mode == VFML_TAA_BILINEAR || mode == VFML_TAA_BILATERAL,
               "vfml_taa_blend: unknown mode %d", mode);
  // the result type is the one numpy's promotion gives the reference (it is the next frame's history)
  const int want = mode == VFML_TAA_SIMPLE ? hist_type : (mode == VFML_TAA_BILINEAR ? VFML_PIX_F32 : VFML_PIX_F64);
  VFML_REQUIRE(out_type == want, "vfml_taa_blend: mode %d with history type %d writes type %d, not %d", mode,
               hist_type, want, out_type);
  VFML_REQUIRE(mode == VFML_TAA_SIMPLE || (flow && (reinterpret_cast<uintptr_t>(flow) & 7u) == 0),
               "vfml_taa_blend: flow missing or not 8-byte aligned");
  VFML_REQUIRE(mode != VFML_TAA_BILATERAL || (h >= 2 && w >= 2), "vfml_taa_blend: bilateral needs a 2x2 image");
  VFML_REQUIRE(out != history, "vfml_taa_blend: in-place history (the gather reads other pixels)");
  TaaArgs a;
  a.cur = current; a.flow = flow; a.hist = history; a.out = out;
  a.h = h; a.w = w;
  a.alpha32 = (float)alpha;
  a.keep64 = 1.0 - alpha;
  a.keep32 = (float)a.keep64;
  const double sigma_sq = sigma_color * sigma_color * 0.1;
  a.denom64 = 2 * sigma_sq + 1e-6;
  a.denom32 = (float)a.denom64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool u8 = cur_type == VFML_PIX_U8, h64 = hist_type == VFML_PIX_F64;
  if (mode == VFML_TAA_SIMPLE) h64 ? launch_taa<VFML_TAA_SIMPLE, double>(a, u8, s) : launch_taa<VFML_TAA_SIMPLE, float>(a, u8, s);
  else if (mode == VFML_TAA_BILINEAR) h64 ? launch_taa<VFML_TAA_BILINEAR, double>(a, u8, s) : launch_taa<VFML_TAA_BILINEAR, float>(a, u8, s);
  else h64 ? launch_taa<VFML_TAA_BILATERAL, double>(a, u8, s) : launch_taa<VFML_TAA_BILATERAL, float>(a, u8, s);
  return vfml_check_launch("vfml_taa_blend");
}

// ---- flow quality map (reference correction_worker.py:175-208) -------------------------------------
// One thread per pixel: warp frame 2 back along the flow (nearest texel by truncation), score the colour match with
// frame 1 (RGB distance, mean absolute difference, cosine similarity), paint green above the threshold, red below,
// full red where the vector leaves the image.  6 B + 8 B read, 3 B written per pixel.  Each float32 step is the single
// IEEE operation torch's CPU kernels perform, fused multiply-adds included where torch has them (bilinear resize of
// LOD fields, vector norms).
namespace {

struct QualityArgs {
  const unsigned char* f1; const unsigned char* f2; const float* flow; unsigned char* out;
  int h, w, fh, fw;
  float scale_y, scale_x;     // fh / h, fw / w (float32 quotients): source step of the bilinear resize
  float mul_x, mul_y;         // float32(w / fw), float32(h / fh): the vector rescale after it
  float threshold;
};

__device__ __forceinline__ float root32(float x) { return (float)sqrt((double)x); }   // correctly rounded
__device__ __forceinline__ float clamp01(float v) { return v != v ? v : fminf(fmaxf(v, 0.0f), 1.0f); }
// float -> int64 -> clamp, with the host's result for NaN and out-of-range values (INT64_MIN, hence 0)
__device__ __forceinline__ int texel(float t, int n) {
  if (!(fabsf(t) < 9.0e18f)) return 0;
  const long long i = (long long)t;
  return (int)(i < 0 ? 0 : (i > n - 1 ? n - 1 : i));
}
// source taps of one output coordinate (align_corners = False)
__device__ __forceinline__ void taps(int dst, float scale, int n_in, int& i0, int& i1, float& l0, float& l1) {
  float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
  src = fmaxf(src, 0.0f);
  i0 = min((int)floorf(src), n_in - 1);
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}

template <bool RESIZE>
__global__ void quality_map_kernel(const QualityArgs a) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)a.h * a.w;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / a.w), x = (int)(p - (int64_t)y * a.w);
    float fx, fy;
    if constexpr (RESIZE) {
      int y0, y1, x0, x1;
      float hy0, hy1, wx0, wx1;
      taps(y, a.scale_y, a.fh, y0, y1, hy0, hy1);
      taps(x, a.scale_x, a.fw, x0, x1, wx0, wx1);
      const float2* fl = (const float2*)a.flow;
      const float2 v00 = fl[(int64_t)y0 * a.fw + x0], v01 = fl[(int64_t)y0 * a.fw + x1];
      const float2 v10 = fl[(int64_t)y1 * a.fw + x0], v11 = fl[(int64_t)y1 * a.fw + x1];
      fx = fmaf(fmaf(v00.x, wx0, v01.x * wx1), hy0, fmaf(v10.x, wx0, v11.x * wx1) * hy1) * a.mul_x;
      fy = fmaf(fmaf(v00.y, wx0, v01.y * wx1), hy0, fmaf(v10.y, wx0, v11.y * wx1) * hy1) * a.mul_y;
    } else {
      const float2 v = ((const float2*)a.flow)[p];
      fx = v.x;
      fy = v.y;
    }
    const float tx = (float)x - fx, ty = (float)y - fy;
    unsigned char r, g;
    if (tx < 0.0f || tx >= (float)a.w || ty < 0.0f || ty >= (float)a.h) {
      r = 255;
      g = 0;
    } else {
      const unsigned char* q1 = a.f1 + 3 * p;
      const unsigned char* q2 = a.f2 + 3 * ((int64_t)texel(ty, a.h) * a.w + texel(tx, a.w));
      float s[3], t[3], d[3];
      for (int c = 0; c < 3; ++c) {
        s[c] = div32((float)q1[c], 255.0f);
        t[c] = div32((float)q2[c], 255.0f);
        d[c] = s[c] - t[c];
      }
      const float rgb = 1.0f - div32(root32((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), 1.732f);
      const float mad = 1.0f - div32((fabsf(d[0]) + fabsf(d[1])) + fabsf(d[2]), 3.0f);
      const float ns = fmaxf(root32(fmaf(s[2], s[2], fmaf(s[1], s[1], s[0] * s[0]))), 1e-8f);
      const float nt = fmaxf(root32(fmaf(t[2], t[2], fmaf(t[1], t[1], t[0] * t[0]))), 1e-8f);
      const float dot = (div32(s[0], ns) * div32(t[0], nt) + div32(s[1], ns) * div32(t[1], nt)) + div32(s[2], ns) * div32(t[2], nt);
      const float cosine = div32(dot + 1.0f, 2.0f);
      const float overall = div32((rgb + mad) + cosine, 3.0f);
      if (overall > a.threshold) {
        r = 0;
        g = (unsigned char)(int)(clamp01((overall - 0.5f) * 2.0f) * 255.0f);
      } else {
        r = (unsigned char)(int)(clamp01(1.0f - overall) * 255.0f);
        g = 0;
      }
    }
    a.out[3 * p] = r;
    a.out[3 * p + 1] = g;
    a.out[3 * p + 2] = 0;
  }
}
}  // namespace

extern "C" int vfml_flow_quality_map(const unsigned char* frame1, const unsigned char* frame2, const float* flow, int fh,
                                     int fw, int h, int w, float threshold, unsigned char* out, void* stream) {
  VFML_REQUIRE(frame1 && frame2 && flow && out && h > 0 && w > 0 && fh > 0 && fw > 0, "vfml_flow_quality_map: bad argument");
  VFML_REQUIRE((reinterpret_cast<uintptr_t>(flow) & 7u) == 0, "vfml_flow_quality_map: flow must be 8-byte aligned");
  VFML_REQUIRE(h <= (1 << 24) && w <= (1 << 24), "vfml_flow_quality_map: image side above 2^24 (float32 pixel grid)");
  QualityArgs a;
  a.f1 = frame1; a.f2 = frame2; a.flow = flow; a.out = out;
  a.h = h; a.w = w; a.fh = fh; a.fw = fw;
  a.scale_y = (float)fh / (float)h;
  a.scale_x = (float)fw / (float)w;
  a.mul_x = (float)((double)w / (double)fw);
  a.mul_y = (float)((double)h / (double)fh);
  a.threshold = threshold;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(blocks_for((int64_t)h * w, 256)), block(256);
  if (fh == h && fw == w) hipLaunchKernelGGL(quality_map_kernel<false>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(quality_map_kernel<true>, grid, block, 0, s, a);
  return vfml_check_launch("vfml_flow_quality_map");
}
