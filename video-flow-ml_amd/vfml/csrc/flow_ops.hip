// Correlation-pyramid lookup (K5), coordinate bookkeeping, 8x convex upsampling (K8).
// All three are HBM/latency-bound gathers: one wave64 per query (lookup) or per coarse pixel
// (upsample), patches staged through LDS, results written as contiguous runs.
#include <stdlib.h>
#include "vfml_common.h"

namespace {

constexpr int MAX_LEVELS = 6;
constexpr int MAX_MAPS = 8;
constexpr int MAX_RADIUS = 4;
constexpr int PATCH = 2 * MAX_RADIUS + 2;  // 10 integer-grid samples per axis
#ifndef VFML_LOOKUP_WAVES
#define VFML_LOOKUP_WAVES 4
#endif
constexpr int LOOKUP_WAVES = VFML_LOOKUP_WAVES;
constexpr int FIXED_LEVELS = 4;   // the fixed-radius kernel's pyramid depth limit (the networks use 4)
static_assert(FIXED_LEVELS == 4, "corr_lookup_fixed_kernel selects among four levels");

struct LookupArgs {
  const float* pyr[MAX_MAPS][MAX_LEVELS];   // one pyramid per query map (problem); rows = that map's queries
  const float* const* table;                // or (non-null) a DEVICE array [map * levels + level] of the same pointers,
                                            // read at run time: the launch can then sit in a captured graph while the
                                            // pyramids it looks up change from replay to replay
  int hl[MAX_LEVELS], wl[MAX_LEVELS], ld[MAX_LEVELS];
  int levels, radius, nq, q_per_map;
  const float* coords; int ld_coords;
  float* out; int ld_out;
  int out16;
  int vol16;     // bit l: level l of the pyramids holds one f16 per element (VFML_FMT_F16 / VFML_VOL_F16_LEVELS; fixed-radius kernels only)
  int tws, ths;  // vol_tile: a level image (and the query grid that orders the rows) stored in (1<<tws) x (1<<ths) tiles; 0, 0: row-major
  int qw;        // width of the query grid (= wl[0]) when tiled: query q reads volume row tiled_at(q / qw, q % qw)
  // two directions in one launch (vfml_corr_lookup_indirect_bidir): maps dir_maps .. 2 dir_maps - 1 are the second direction
  // of the same queries - their coordinates dir_coords floats, their output dir_out floats behind the first direction's
  int dir_maps, dir_coords, dir_out;
  int dir_tab;   // table index (in maps) of the second direction's first pyramid
};

// query q of a launch -> the row of coords / out it belongs to and the float offsets of its direction
__device__ __forceinline__ void lookup_rows(const LookupArgs& a, int q, int& map, int64_t& crow, int64_t& orow) {
  const int dir = a.dir_maps && map >= a.dir_maps;
  const int64_t qd = q - (dir ? a.dir_maps * a.q_per_map : 0);
  if (dir) map += a.dir_tab - a.dir_maps;        // (from here on `map` names the pyramid: its place in the table)
  crow = qd * a.ld_coords + (dir ? a.dir_coords : 0);
  orow = qd * a.ld_out + (dir ? a.dir_out : 0);
}

// Position of texel (y, x) of a w-wide image stored as (1<<tws) x (1<<ths) tiles, tile after tile, each tile row-major
// (tws = ths = 0: plain row-major).  A (2r+2)^2 lookup window then lies in ~8 128-byte lines instead of ~13 (ten 40-byte row
// segments): the lookup is bound by the lines it drags in, not by the texels it uses (tools/exp/lookup_tiled.py).
__device__ __forceinline__ int tiled_at(int y, int x, int w, int tws, int ths) {
  const int tpr = (w + (1 << tws) - 1) >> tws;
  return ((((y >> ths) * tpr + (x >> tws)) << (tws + ths)) + ((y & ((1 << ths) - 1)) << tws) + (x & ((1 << tws) - 1)));
}

// grid_sample's bilinear mix nw*(1-fx)(1-fy) + ne*fx(1-fy) + sw*(1-fx)fy + se*fx*fy as ONE explicit chain of fused
// multiply-adds, shared by both lookup kernels so that they agree bit for bit whatever hipcc would contract on its own.
// It also keeps the mix out of v_pk_mul_f32: left to the vectoriser, the fixed-radius kernel multiplied the (sw, se) pair
// as a packed op straight after the s_waitcnt of its ds_read2_b32, and with one of this library's MFMA kernels on a second
// stream that packed multiply read a stale `sw` register in lanes 48-63 a few dozen times per launch
// (profiles/r02_kernel_anatomy.md section 7, tools/exp/two_stream_lookup_diag.py).
__device__ __forceinline__ float bilinear4(float nw, float ne, float sw, float se, float wx0, float fx, float wy0, float fy) {
  float v = ne * (fx * wy0);
  v = __builtin_fmaf(nw, wx0 * wy0, v);
  v = __builtin_fmaf(sw, wx0 * fy, v);
  v = __builtin_fmaf(se, fx * fy, v);
  return v;
}

// One wave per query.  Per level the window's (2r+1)^2 bilinear samples all share the same
// fractional offset, so they are blends of one (2r+2)^2 integer-grid patch: the wave gathers the
// patch (zero outside the level) into LDS, then each lane produces output channels
// o = l*(2r+1)^2 + i*(2r+1) + j  (x + d[i], y + d[j]: RAFT's window order), written contiguously.
__global__ __launch_bounds__(64 * LOOKUP_WAVES) void corr_lookup_kernel(const LookupArgs a) {
  __shared__ float patch[LOOKUP_WAVES][MAX_LEVELS][PATCH * PATCH];
  __shared__ float frac[LOOKUP_WAVES][MAX_LEVELS][2];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int q = blockIdx.x * LOOKUP_WAVES + wv;
  const bool live = q < a.nq;
  int map = live ? q / a.q_per_map : 0;
  const int qq = q - map * a.q_per_map;      // row inside that map's pyramid
  const int side = 2 * a.radius + 2;  // patch side
  const int win = 2 * a.radius + 1;
  const int psz = side * side;
  int64_t crow = 0, orow = 0;
  if (live) lookup_rows(a, q, map, crow, orow);
  if (live) {
    const float cx = a.coords[crow + 0];
    const float cy = a.coords[crow + 1];
    const int total = a.levels * psz;
    const int qrow = (a.tws | a.ths) ? tiled_at(qq / a.qw, qq % a.qw, a.qw, a.tws, a.ths) : qq;
    for (int e = lane; e < total; e += 64) {
      const int l = e / psz;
      const int idx = e - l * psz;
      const int py = idx / side, px = idx - py * side;
      const float inv = 1.0f / (float)(1 << l);
      const float x = cx * inv, y = cy * inv;  // exact: power-of-two scale
      const float fx0 = floorf(x), fy0 = floorf(y);
      // clamp before the int conversion so that wild coordinates cannot overflow
      const int x0 = (int)fminf(fmaxf(fx0, -65536.f), 65536.f) - a.radius;
      const int y0 = (int)fminf(fmaxf(fy0, -65536.f), 65536.f) - a.radius;
      const int xx = x0 + px, yy = y0 + py;
      float v = 0.f;
      if (xx >= 0 && xx < a.wl[l] && yy >= 0 && yy < a.hl[l])
        v = (a.table ? a.table[map * a.levels + l] : a.pyr[map][l])[(int64_t)qrow * a.ld[l] + tiled_at(yy, xx, a.wl[l], a.tws, a.ths)];
      patch[wv][l][idx] = v;
      if (idx == 0) {
        frac[wv][l][0] = x - fx0;
        frac[wv][l][1] = y - fy0;
      }
    }
  }
  __syncthreads();
  if (!live) return;
  const int nout = a.levels * win * win;
  float* o = a.out + orow;
  const int nwrite = a.out16 ? (nout + 7) & ~7 : nout;   // split rows: zero-fill up to a whole unit
  for (int c = lane; c < nwrite; c += 64) {
    if (c >= nout) {
      _Float16* u = reinterpret_cast<_Float16*>(o + (c & ~7));
      u[c & 7] = (_Float16)0.f;
      u[8 + (c & 7)] = (_Float16)0.f;
      continue;
    }
    const int l = c / (win * win);
    const int rem = c - l * win * win;
    const int i = rem / win, j = rem - i * win;  // i: x offset index, j: y offset index
    const float fx = frac[wv][l][0], fy = frac[wv][l][1];
    const float* p = &patch[wv][l][j * side + i];
    // grid_sample's bilinear: nw*(1-fx)(1-fy) + ne*fx(1-fy) + sw*(1-fx)fy + se*fx*fy
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    const float v = bilinear4(p[0], p[1], p[side], p[side + 1], wx0, fx, wy0, fy);
    if (a.out16) {
      vfml_h16x2 hh, ll;
      vfml_split2(v, 0.f, hh, ll);
      _Float16* u = reinterpret_cast<_Float16*>(o + (c & ~7));
      u[c & 7] = hh[0];
      u[8 + (c & 7)] = ll[0];
    } else {
      o[c] = v;
    }
  }
}

// The bilinear mix with the four weight products already formed (the same products, the same chain: bit-identical to bilinear4).
__device__ __forceinline__ float bilinear4w(float nw, float ne, float sw, float se, float w00, float w10, float w01, float w11) {
  float v = ne * w10;
  v = __builtin_fmaf(nw, w00, v);
  v = __builtin_fmaf(sw, w01, v);
  v = __builtin_fmaf(se, w11, v);
  return v;
}

// What one wave of the fixed-radius kernel orders between its LDS writes and the reads other lanes make of them (LDS
// operations of a wave execute in order; this keeps the compiler from moving them and waits for the counters).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The same lookup with the radius as a compile-time constant (4: default, 3: --fast), one wave per query, no workgroup
// barrier.  The kernel is bound by the instructions it issues as much as by the lines it gathers (round 2: ~710 VALU
// instructions per query = 1.5 us of a SIMD, 95 queries per SIMD per launch at 1080p), so every phase is laid out for few
// instructions:
//   gather    level by level (compile-time level: shapes, row pointer and window origin sit in scalar registers, the
//             power-of-two scale is a constant, no per-lane selects), a level's (2R+2)^2 texels in two passes of the wave;
//             every load unconditional on a clamped address, the value zeroed afterwards if the texel is outside
//   mix       lane (level, x offset) walks its column of 2R+1 samples downwards: the lower texel pair of one sample is the
//             upper pair of the next (one ds_read2 and four multiply-adds per sample, no index arithmetic)
//   store     the samples go through LDS so that a lane owns one 8-channel unit: 32 bytes of split row (or two quads of f32)
//   V16       bit l set: level l of the volumes holds f16 texels (a compile-time property of each unrolled level)
template <int R, bool OUT16, int V16 = 0>
__global__ __launch_bounds__(64 * LOOKUP_WAVES) void corr_lookup_fixed_kernel(const LookupArgs a) {
  constexpr int SIDE = 2 * R + 2, PSZ = SIDE * SIDE, WIN = 2 * R + 1, WW = WIN * WIN;
  constexpr int NOUTPAD = (FIXED_LEVELS * WW + 7) & ~7;
  static_assert(PSZ <= 128 && FIXED_LEVELS * WIN <= 64 && NOUTPAD / 8 <= 64, "one wave: at most two gather passes per level");
  __shared__ float patch[LOOKUP_WAVES][FIXED_LEVELS][PSZ];
  __shared__ __attribute__((aligned(16))) float outs[LOOKUP_WAVES][NOUTPAD];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x * LOOKUP_WAVES + wv;       // the wave's query: everything derived from it is scalar
  if (q >= a.nq) return;
  // (integer division runs on the vector unit: readfirstlane tells the compiler its result is still one value per wave)
  int map = __builtin_amdgcn_readfirstlane(q / a.q_per_map);
  const int qq = q - map * a.q_per_map;
  const int qrow = __builtin_amdgcn_readfirstlane((a.tws | a.ths) ? tiled_at(qq / a.qw, qq % a.qw, a.qw, a.tws, a.ths) : qq);
  int64_t crow, orow;
  lookup_rows(a, q, map, crow, orow);
  const float cx = a.coords[crow + 0];
  const float cy = a.coords[crow + 1];
  // the two passes of a level: window cells lane and lane + 64 (the second pass: cells 64 .. PSZ-1)
  const int pyA = lane / SIDE, pxA = lane - pyA * SIDE;
  const int pyB = (lane + 64) / SIDE, pxB = (lane + 64) - pyB * SIDE;
  const bool inB = lane + 64 < PSZ;
  // Row pointers of all levels first (a device table is read here), then STRAIGHT-LINE code for the gathers: behind a
  // branch - even a uniform `l < levels` - hipcc waits for a level's texels before it asks for the next level's.  A level the
  // call does not have reads level 0's first texel and drops it.
  const float* pl[FIXED_LEVELS];
  if (a.table) {
#pragma unroll
    for (int l = 0; l < FIXED_LEVELS; ++l) pl[l] = a.table[map * a.levels + (l < a.levels ? l : 0)];
  } else {
#pragma unroll
    for (int l = 0; l < FIXED_LEVELS; ++l) pl[l] = a.pyr[map][l < a.levels ? l : 0];
  }
  float va[FIXED_LEVELS], vb[FIXED_LEVELS];
  bool oka[FIXED_LEVELS], okb[FIXED_LEVELS];
#pragma unroll
  for (int l = 0; l < FIXED_LEVELS; ++l) {
    const bool have = l < a.levels;
    const int ll = have ? l : 0;
    const bool v16 = (V16 >> l) & 1;         // (l is unrolled: a constant per copy of the body)
    const int ES = v16 ? 2 : 4;              // bytes per texel of this level
    const char* base = reinterpret_cast<const char*>(pl[l]) + (int64_t)qrow * a.ld[ll] * ES;
    const int wl = a.wl[ll], hl = a.hl[ll];
    const float inv = __builtin_bit_cast(float, (127 - l) << 23);      // 2^-l, exactly what 1.0f / (1 << l) is
    const float x = cx * inv, y = cy * inv;                            // exact: power-of-two scale
    const int x0 = (int)fminf(fmaxf(floorf(x), -65536.f), 65536.f) - R;   // (clamped before the conversion: wild coordinates)
    const int y0 = (int)fminf(fmaxf(floorf(y), -65536.f), 65536.f) - R;
    auto texel = [&](int px, int py, bool in, float& out, bool& ok) {
      const int xx = x0 + px, yy = y0 + py;
      ok = have && in && xx >= 0 && xx < wl && yy >= 0 && yy < hl;
      const int at = tiled_at(min(max(yy, 0), hl - 1), min(max(xx, 0), wl - 1), wl, a.tws, a.ths);
      // (explicitly GLOBAL: a pointer that went through selects or a table loses its address space and becomes flat_load)
      if (v16) out = (float)((const __attribute__((address_space(1))) _Float16*)base)[at];
      else out = ((const __attribute__((address_space(1))) float*)base)[at];          // (zeroed below, once every load is on its way)
    };
    texel(pxA, pyA, true, va[l], oka[l]);
    if constexpr (PSZ > 64) texel(pxB, pyB, inB, vb[l], okb[l]);
    else { vb[l] = 0.f; okb[l] = false; }
  }
#pragma unroll
  for (int l = 0; l < FIXED_LEVELS; ++l) {
    va[l] = oka[l] ? va[l] : 0.f;
    vb[l] = okb[l] ? vb[l] : 0.f;
  }
#pragma unroll
  for (int l = 0; l < FIXED_LEVELS; ++l) {
    if (l < a.levels) {
      if (PSZ > 64 || lane < PSZ) patch[wv][l][lane] = va[l];
      if constexpr (PSZ > 64) if (inB) patch[wv][l][lane + 64] = vb[l];
    }
  }
  const int nout = a.levels * WW;
  const int nunits = (nout + 7) >> 3;
  if (lane >= 56 && nout + (lane - 56) < nunits * 8) outs[wv][nout + (lane - 56)] = 0.f;   // the zero channels that fill the last unit
  wave_lds_sync();
  if (lane < a.levels * WIN) {
    const int l = lane / WIN, i = lane - l * WIN;          // i: x offset index; the lane walks j, the y offset index
    const float inv = __builtin_bit_cast(float, (127 - l) << 23);
    const float x = cx * inv, y = cy * inv;
    const float fx = x - floorf(x), fy = y - floorf(y);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    const float w00 = wx0 * wy0, w10 = fx * wy0, w01 = wx0 * fy, w11 = fx * fy;
    const float* p = &patch[wv][l][i];
    float* o = &outs[wv][l * WW + i * WIN];
    float nw = p[0], ne = p[1];
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const float sw = p[(j + 1) * SIDE], se = p[(j + 1) * SIDE + 1];
      o[j] = bilinear4w(nw, ne, sw, se, w00, w10, w01, w11);
      nw = sw; ne = se;
    }
  }
  wave_lds_sync();
  if (lane < nunits) {
    float* o = a.out + orow + lane * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(&outs[wv][lane * 8]);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(&outs[wv][lane * 8 + 4]);
    if (OUT16) {
      // one unit of the split row: eight hi halves, then the eight lo halves
      vfml_h16x2 h0, h1, h2, h3, l0, l1, l2, l3;
      vfml_split2(v0[0], v0[1], h0, l0);
      vfml_split2(v0[2], v0[3], h1, l1);
      vfml_split2(v1[0], v1[1], h2, l2);
      vfml_split2(v1[2], v1[3], h3, l3);
      uint4 hv, lv;
      hv.x = __builtin_bit_cast(unsigned, h0); hv.y = __builtin_bit_cast(unsigned, h1);
      hv.z = __builtin_bit_cast(unsigned, h2); hv.w = __builtin_bit_cast(unsigned, h3);
      lv.x = __builtin_bit_cast(unsigned, l0); lv.y = __builtin_bit_cast(unsigned, l1);
      lv.z = __builtin_bit_cast(unsigned, l2); lv.w = __builtin_bit_cast(unsigned, l3);
      *reinterpret_cast<uint4*>(o) = hv;
      *reinterpret_cast<uint4*>(o + 4) = lv;
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = lane * 8 + 4 * h;
        const f32x4 w = h ? v1 : v0;
        if (c + 3 < nout && (((uintptr_t)(o + 4 * h)) & 15u) == 0) {
          *reinterpret_cast<f32x4*>(o + 4 * h) = w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c + k < nout) o[4 * h + k] = w[k];
        }
      }
    }
  }
}

__global__ void coords_init_kernel(f32x4* __restrict__ coords, int h, int w, int64_t total) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % w);
    const int y = (int)((p / w) % h);
    f32x4 v = {(float)x, (float)y, (float)x, (float)y};
    coords[p] = v;
  }
}

// rows[p] = the seven horizontal taps' flow quads of pixel p as 32 split-row channels (include/vfml.h vfml_flow_rows7):
// one thread per (pixel, 8-channel unit) = two taps
__global__ void flow_rows7_kernel(const f32x4* __restrict__ flow, int w, int64_t total, char* __restrict__ rows) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total * 4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i >> 2;
    const int u = (int)(i & 3);
    const int x = (int)(p % w);
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
    const int x0 = x + 2 * u - 3, x1 = x0 + 1;
    if (x0 >= 0 && x0 < w) t0 = flow[p + 2 * u - 3];
    if (u < 3 && x1 >= 0 && x1 < w) t1 = flow[p + 2 * u - 2];        // (unit 3's second quad = channels 28..31: zero)
    vfml_h16x2 h[4], l[4];
    vfml_split2(t0[0], t0[1], h[0], l[0]);
    vfml_split2(t0[2], t0[3], h[1], l[1]);
    vfml_split2(t1[0], t1[1], h[2], l[2]);
    vfml_split2(t1[2], t1[3], h[3], l[3]);
    uint4 hv, lv;
    hv.x = __builtin_bit_cast(unsigned, h[0]); hv.y = __builtin_bit_cast(unsigned, h[1]);
    hv.z = __builtin_bit_cast(unsigned, h[2]); hv.w = __builtin_bit_cast(unsigned, h[3]);
    lv.x = __builtin_bit_cast(unsigned, l[0]); lv.y = __builtin_bit_cast(unsigned, l[1]);
    lv.z = __builtin_bit_cast(unsigned, l[2]); lv.w = __builtin_bit_cast(unsigned, l[3]);
    char* o = rows + p * 128 + u * 32;
    *reinterpret_cast<uint4*>(o) = hv;
    *reinterpret_cast<uint4*>(o + 16) = lv;
  }
}

// coords[p] (already updated) -> flow = coords - grid, written to fa[p*lda..+4] (f32) and fb[p*ldb..+4] (f32, or the second quad
// of a split-row unit)
__device__ __forceinline__ void emit_flow(int64_t p, const f32x4 c, int h, int w, float* __restrict__ fa, int lda,
                                          float* __restrict__ fb, int ldb, int b16) {
  const float x = (float)(int)(p % w);
  const float y = (float)(int)((p / w) % h);
  const f32x4 f = {c[0] - x, c[1] - y, c[2] - x, c[3] - y};
  if (fa) *reinterpret_cast<f32x4*>(fa + p * lda) = f;
  if (fb) {
    if (b16) {   // second quad of a split-row unit: hi halves at +0 (8 B), lo halves 16 B further
      typedef vfml_h16x2 fp16x2_;
      fp16x2_ h0, h1, l0, l1;
      vfml_split2(f[0], f[1], h0, l0);
      vfml_split2(f[2], f[3], h1, l1);
      // fb points at channel 4 of the unit = byte 16 of it in f32 addressing; the quad slot of the
      // hi halves is byte 8 of the unit
      char* u = reinterpret_cast<char*>(fb + p * ldb) - 16;
      *reinterpret_cast<fp16x2_*>(u + 8) = h0;
      *reinterpret_cast<fp16x2_*>(u + 12) = h1;
      *reinterpret_cast<fp16x2_*>(u + 24) = l0;
      *reinterpret_cast<fp16x2_*>(u + 28) = l1;
    } else {
      *reinterpret_cast<f32x4*>(fb + p * ldb) = f;
    }
  }
}

// out[p] = bias + sum of the nine taps' quads of the tap-major 36-column map t (include/vfml.h vfml_tapsum3x3); with `coords`
// the sum is the flow update of the iteration: coords[p] += it and the flows are emitted as vfml_coords_update does
// (vfml_tapsum3x3_update: one launch instead of two)
__global__ void tapsum3x3_kernel(const float* __restrict__ t, int ld, const float* __restrict__ bias, int h, int w,
                                 int64_t total, f32x4* __restrict__ out, int parts, int64_t part_stride,
                                 f32x4* __restrict__ coords, float* __restrict__ fa, int lda, float* __restrict__ fb, int ldb,
                                 int b16) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(p % w), y = (int)((p / w) % h);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (bias) s = f32x4{bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
          const float* q = t + (p + (int64_t)(ky - 1) * w + (kx - 1)) * ld + (ky * 3 + kx) * 4;
          f32x4 v = *reinterpret_cast<const f32x4*>(q);
          for (int k = 1; k < parts; ++k) v = v + *reinterpret_cast<const f32x4*>(q + k * part_stride);
          s = s + v;
        }
      }
    if (out) out[p] = s;
    if (coords) {
      const f32x4 c = coords[p] + s;
      coords[p] = c;
      emit_flow(p, c, h, w, fa, lda, fb, ldb, b16);
    }
  }
}

__global__ void coords_update_kernel(f32x4* __restrict__ coords, const f32x4* __restrict__ delta, int h, int w,
                                     int64_t total, float* __restrict__ fa, int lda, float* __restrict__ fb, int ldb,
                                     int b16) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    f32x4 c = coords[p];
    if (delta) {
      c = c + delta[p];
      coords[p] = c;
    }
    emit_flow(p, c, h, w, fa, lda, fb, ldb, b16);
  }
}

// One wave per coarse pixel, lane = sub-pixel (sy*8+sx).  9 coalesced 256-B mask reads, softmax
// over the 9 taps in registers, 3x3 coarse flow neighbourhood (zero outside, x8), float2 stores.
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ coords, int ch,
                                                             const float* __restrict__ mask, int ld_mask, int h,
                                                             int w, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= h * w) return;
  const int y = p / w, x = p - y * w;
  const float* m = mask + (int64_t)p * ld_mask + lane;
  float logit[9], fxv[9], fyv[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    logit[k] = m[k * 64];
    mx = fmaxf(mx, logit[k]);
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float u = 0.f, v = 0.f;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
      const float* c = coords + ((int64_t)yy * w + xx) * 4 + ch;
      u = 8.f * (c[0] - (float)xx);
      v = 8.f * (c[1] - (float)yy);
    }
    fxv[k] = u;
    fyv[k] = v;
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    logit[k] = expf(logit[k] - mx);
    s += logit[k];
  }
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float pk = logit[k] / s;
    ax += pk * fxv[k];
    ay += pk * fyv[k];
  }
  const int sy = lane >> 3, sx = lane & 7;
  float2* o = reinterpret_cast<float2*>(out) + ((int64_t)(8 * y + sy) * (8 * w) + 8 * x + sx);
  *o = make_float2(ax, ay);
}

__global__ void flow_lod_kernel(const float2* __restrict__ flow, int h, int w, int ho, int wo, float2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)ho * wo;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(i / wo), ox = (int)(i - (int64_t)oy * wo);
    const int y0 = 2 * oy, x0 = 2 * ox;
    const bool y1 = y0 + 1 < h, x1 = x0 + 1 < w;
    // row-major over the block, missing cells contribute 0 with weight 0 (as np.sum(block * weights))
    const float2 a = flow[(int64_t)y0 * w + x0];
    const float2 b = x1 ? flow[(int64_t)y0 * w + x0 + 1] : make_float2(0.f, 0.f);
    const float2 c = y1 ? flow[(int64_t)(y0 + 1) * w + x0] : make_float2(0.f, 0.f);
    const float2 d = (y1 && x1) ? flow[(int64_t)(y0 + 1) * w + x0 + 1] : make_float2(0.f, 0.f);
    const float wt = ((1.f + (x1 ? 1.f : 0.f)) + (y1 ? 1.f : 0.f)) + ((y1 && x1) ? 1.f : 0.f);
    const float sx = ((a.x + b.x) + c.x) + d.x, sy = ((a.y + b.y) + c.y) + d.y;
    out[i] = make_float2(sx / wt * 0.5f, sy / wt * 0.5f);
  }
}

inline int grid_for(int64_t items, int block) {
  int64_t g = (items + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// n device pointers -> a device table (vfml_corr_lookup_indirect reads it): the values travel as kernel arguments, so
// no host staging buffer has to outlive the call
constexpr int MAX_TABLE = MAX_MAPS * MAX_LEVELS;
struct PtrTable { const void* p[MAX_TABLE]; };
__global__ void ptr_table_kernel(const PtrTable t, int n, const void** dst) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = t.p[threadIdx.x];
}

}  // namespace

extern "C" int vfml_ptr_table_set(void* table, const void* const* ptrs, int n, void* stream) {
  VFML_REQUIRE(table && ptrs && n >= 1 && n <= MAX_TABLE, "vfml_ptr_table_set: 1..%d pointers", MAX_TABLE);
  PtrTable t;
  for (int i = 0; i < MAX_TABLE; ++i) t.p[i] = i < n ? ptrs[i] : nullptr;
  hipLaunchKernelGGL(ptr_table_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), t, n,
                     reinterpret_cast<const void**>(table));
  return vfml_check_launch("vfml_ptr_table_set");
}

static int corr_lookup_impl(const float* const* pyr, const float* const* table, const int32_t* hl, const int32_t* wl,
                            const int32_t* ld, int levels, int radius, int nmaps, int q_per_map, const float* coords,
                            int ld_coords, float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile, void* stream,
                            int dir_maps = 0, int dir_coords = 0, int dir_out = 0, int dir_tab = 0);

extern "C" int vfml_corr_lookup(const float* const* pyr, const int32_t* hl, const int32_t* wl, const int32_t* ld,
                                int levels, int radius, int nmaps, int q_per_map, const float* coords, int ld_coords,
                                float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile, void* stream) {
  VFML_REQUIRE(pyr, "vfml_corr_lookup: null pointer");
  return corr_lookup_impl(pyr, nullptr, hl, wl, ld, levels, radius, nmaps, q_per_map, coords, ld_coords, out, ld_out, out_fmt,
                          vol_fmt, vol_tile, stream);
}

extern "C" int vfml_corr_lookup_indirect(const float* const* table, const int32_t* hl, const int32_t* wl, const int32_t* ld,
                                         int levels, int radius, int nmaps, int q_per_map, const float* coords,
                                         int ld_coords, float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile, void* stream) {
  VFML_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7u) == 0, "vfml_corr_lookup_indirect: null / misaligned table");
  return corr_lookup_impl(nullptr, table, hl, wl, ld, levels, radius, nmaps, q_per_map, coords, ld_coords, out, ld_out,
                          out_fmt, vol_fmt, vol_tile, stream);
}

extern "C" int vfml_corr_lookup_indirect_bidir(const float* const* table, const int32_t* hl, const int32_t* wl, const int32_t* ld,
                                               int levels, int radius, int nmaps, int q_per_map, const float* coords,
                                               int ld_coords, int dir_coords, float* out, int ld_out, int dir_out, int dir_tab,
                                               int out_fmt, int vol_fmt, int vol_tile, void* stream) {
  VFML_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7u) == 0, "vfml_corr_lookup_indirect_bidir: null / misaligned table");
  VFML_REQUIRE(nmaps >= 1 && 2 * nmaps <= MAX_MAPS, "vfml_corr_lookup_indirect_bidir: nmaps=%d out of [1,%d]", nmaps, MAX_MAPS / 2);
  VFML_REQUIRE(dir_coords >= 0 && dir_coords + 2 <= ld_coords && dir_out >= 0 && (out_fmt != VFML_FMT_S16 || dir_out % 8 == 0),
               "vfml_corr_lookup_indirect_bidir: dir_coords within a coords row, dir_out >= 0 (a multiple of 8 floats for split rows)");
  VFML_REQUIRE(dir_tab >= nmaps && (dir_tab + nmaps) * levels <= MAX_TABLE,
               "vfml_corr_lookup_indirect_bidir: dir_tab=%d (the second direction's first map in the table) out of range", dir_tab);
  return corr_lookup_impl(nullptr, table, hl, wl, ld, levels, radius, 2 * nmaps, q_per_map, coords, ld_coords, out, ld_out,
                          out_fmt, vol_fmt, vol_tile, stream, nmaps, dir_coords, dir_out, dir_tab);
}

static int corr_lookup_impl(const float* const* pyr, const float* const* table, const int32_t* hl, const int32_t* wl,
                            const int32_t* ld, int levels, int radius, int nmaps, int q_per_map, const float* coords,
                            int ld_coords, float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile, void* stream,
                            int dir_maps, int dir_coords, int dir_out, int dir_tab) {
  VFML_REQUIRE(nmaps >= 1 && nmaps <= MAX_MAPS && q_per_map > 0, "vfml_corr_lookup: nmaps=%d out of [1,%d] or empty maps", nmaps, MAX_MAPS);
  const int nq = nmaps * q_per_map;
  VFML_REQUIRE(out_fmt == VFML_FMT_F32 || out_fmt == VFML_FMT_S16, "vfml_corr_lookup: bad out_fmt");
  // f16 levels as a mask: VFML_FMT_F16 = every level; VFML_VOL_F16_LEVELS(m) = the levels of m (the forms built: all, all
  // but level 0, levels 2-3, level 3)
  int v16mask = vol_fmt == VFML_FMT_F16 ? 15 : ((vol_fmt & ~15) == VFML_VOL_F16_LEVELS(0) ? (vol_fmt & 15) : (vol_fmt == VFML_FMT_F32 ? 0 : -1));
  if (v16mask > 0 && levels >= 1 && levels <= FIXED_LEVELS) {
    v16mask &= (1 << levels) - 1;                                  // levels the pyramid does not have do not count ...
    if (v16mask) v16mask |= 15 & ~((1 << levels) - 1);             // ... and take whatever form is built
  }
  VFML_REQUIRE(v16mask == 0 || ((radius == 3 || radius == 4) && levels <= FIXED_LEVELS &&
                                (v16mask == 15 || v16mask == 14 || v16mask == 12 || v16mask == 8)),
               "vfml_corr_lookup: vol_fmt is VFML_FMT_F32, or (radius 3 / 4, at most %d levels) VFML_FMT_F16 or "
               "VFML_VOL_F16_LEVELS(m) with m = levels 1-3, 2-3 or 3", FIXED_LEVELS);
  VFML_REQUIRE(hl && wl && ld && coords && out, "vfml_corr_lookup: null pointer");
  VFML_REQUIRE(levels >= 1 && levels <= MAX_LEVELS, "vfml_corr_lookup: levels=%d out of [1,%d]", levels, MAX_LEVELS);
  VFML_REQUIRE(radius >= 1 && radius <= MAX_RADIUS, "vfml_corr_lookup: radius=%d out of [1,%d]", radius, MAX_RADIUS);
  VFML_REQUIRE(ld_coords >= 2, "vfml_corr_lookup: bad ld_coords");
  const int nout = levels * (2 * radius + 1) * (2 * radius + 1);
  VFML_REQUIRE(ld_out >= nout, "vfml_corr_lookup: ld_out=%d < %d channels", ld_out, nout);
  if (out_fmt == VFML_FMT_S16)
    VFML_REQUIRE(ld_out % 8 == 0 && ld_out >= ((nout + 7) & ~7) && (reinterpret_cast<uintptr_t>(out) & 31u) == 0,
                 "vfml_corr_lookup: split-row output needs a 32-byte aligned out and ld_out %% 8 == 0");
  LookupArgs a;
  a.out16 = out_fmt == VFML_FMT_S16;
  a.vol16 = v16mask;
  a.table = table;
  for (int m = 0; m < MAX_MAPS; ++m)
    for (int l = 0; l < MAX_LEVELS; ++l) a.pyr[m][l] = nullptr;
  for (int l = 0; l < levels; ++l) {
    VFML_REQUIRE(hl[l] > 0 && wl[l] > 0 && ld[l] >= hl[l] * wl[l], "vfml_corr_lookup: bad level %d", l);
    a.hl[l] = hl[l]; a.wl[l] = wl[l]; a.ld[l] = ld[l];
    for (int m = 0; m < nmaps && pyr; ++m) {
      VFML_REQUIRE(pyr[m * levels + l], "vfml_corr_lookup: null pyramid pointer (map %d, level %d)", m, l);
      a.pyr[m][l] = pyr[m * levels + l];
    }
  }
  for (int l = levels; l < MAX_LEVELS; ++l) a.hl[l] = a.wl[l] = a.ld[l] = 0;
  a.tws = vol_tile & 15; a.ths = (vol_tile >> 4) & 15; a.qw = wl[0];
  VFML_REQUIRE(vol_tile >= 0 && vol_tile < 256 && a.tws <= 6 && a.ths <= 6, "vfml_corr_lookup: vol_tile = tws + 16 * ths with tws, ths <= 6");
  if (vol_tile) {
    VFML_REQUIRE(q_per_map == hl[0] * wl[0], "vfml_corr_lookup: a tiled volume orders its rows by the level-0 grid (q_per_map == hl[0] * wl[0])");
    for (int l = 0; l < levels; ++l)
      VFML_REQUIRE((int64_t)ld[l] >= (int64_t)(((hl[l] - 1) >> a.ths) + 1) * (((wl[l] - 1) >> a.tws) + 1) << (a.tws + a.ths),
                   "vfml_corr_lookup: ld[%d]=%d is less than the whole tiles of a %d x %d level", l, ld[l], wl[l], hl[l]);
  }
  a.levels = levels; a.radius = radius; a.nq = nq; a.q_per_map = q_per_map;
  a.coords = coords; a.ld_coords = ld_coords; a.out = out; a.ld_out = ld_out;
  a.dir_maps = dir_maps; a.dir_coords = dir_coords; a.dir_out = dir_out; a.dir_tab = dir_tab;
  const dim3 grid((nq + LOOKUP_WAVES - 1) / LOOKUP_WAVES), block(64 * LOOKUP_WAVES);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const int generic = getenv("VFML_LOOKUP_GENERIC") ? atoi(getenv("VFML_LOOKUP_GENERIC")) : 0;
  if (a.vol16) {
    const int m = a.vol16;
#define VFML_LOOKUP16(RR, OO)                                                                                  \
  do {                                                                                                          \
    if (m == 15) hipLaunchKernelGGL((corr_lookup_fixed_kernel<RR, OO, 15>), grid, block, 0, st, a);             \
    else if (m == 14) hipLaunchKernelGGL((corr_lookup_fixed_kernel<RR, OO, 14>), grid, block, 0, st, a);        \
    else if (m == 12) hipLaunchKernelGGL((corr_lookup_fixed_kernel<RR, OO, 12>), grid, block, 0, st, a);        \
    else hipLaunchKernelGGL((corr_lookup_fixed_kernel<RR, OO, 8>), grid, block, 0, st, a);                      \
  } while (0)
    if (radius == 4) {
      if (a.out16) VFML_LOOKUP16(4, true);
      else VFML_LOOKUP16(4, false);
    } else {
      if (a.out16) VFML_LOOKUP16(3, true);
      else VFML_LOOKUP16(3, false);
    }
#undef VFML_LOOKUP16
  } else if (radius == 4 && !generic && levels <= FIXED_LEVELS) {
    if (a.out16) hipLaunchKernelGGL((corr_lookup_fixed_kernel<4, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((corr_lookup_fixed_kernel<4, false>), grid, block, 0, st, a);
  } else if (radius == 3 && !generic && levels <= FIXED_LEVELS) {
    if (a.out16) hipLaunchKernelGGL((corr_lookup_fixed_kernel<3, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((corr_lookup_fixed_kernel<3, false>), grid, block, 0, st, a);
  } else {
    hipLaunchKernelGGL(corr_lookup_kernel, grid, block, 0, st, a);
  }
  return vfml_check_launch("vfml_corr_lookup");
}

// ---- flow -> 8-bit motion-vector images (reference encoding/flow_encoders.py) ---------------------------
// Byte outputs: every float32 operation of the reference's numpy code is done as the same single
// IEEE operation (correctly rounded divide and square root, no fused multiply-add), NaN / inf flow
// through clip / nan_to_num exactly as numpy's minimum(maximum()) and astype(uint8) treat them.
namespace {
#pragma clang fp contract(off)
__device__ __forceinline__ float np_clip(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }
// float32 square root, correctly rounded like numpy's: through the double-precision root (53 >= 2*24 + 2 bits:
// the second rounding is innocuous); v_sqrt_f32 alone is 1 ulp off on ~2e-5 of the inputs
__device__ __forceinline__ float np_sqrt(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ unsigned char np_u8(float v) {      // nan_to_num(nan=0, posinf=255, neginf=0).astype(uint8)
  if (v != v) return 0;
  if (v == INFINITY) return 255;
  if (v == -INFINITY) return 0;
  return (unsigned char)(int)v;
}

struct EncodeArgs {
  const float* flow; unsigned char* out; int64_t n;
  int mode;                        // VFML_ENCODE_*
  float width, height, scale;      // gamedev: divide by the image size, then scale
  float clamp, two_clamp;          // float32(clamp_range), float32(2 * clamp_range)
};

__global__ void flow_encode_kernel(const EncodeArgs a) {
#pragma clang fp contract(off)
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * blockDim.x) {
    float fx = a.flow[2 * p], fy = a.flow[2 * p + 1];
    unsigned char r, g, b;
    if (a.mode == VFML_ENCODE_RGB8) {
      float dx = fx / a.clamp, dy = fy / a.clamp;
      const float len = np_sqrt(dx * dx + dy * dy);
      if (len > 1.0f) {
        dx = dx / len;
        dy = dy / len;
      }
      const float corr = np_sqrt((1.0f - dx * dx) - dy * dy);
      r = np_u8(((np_clip(dx, -1.0f, 1.0f) + 1.0f) / 2.0f) * 255.0f);
      g = np_u8(((np_clip(dy, -1.0f, 1.0f) + 1.0f) / 2.0f) * 255.0f);
      b = np_u8(corr * 255.0f);
    } else {
      if (a.mode == VFML_ENCODE_GAMEDEV) {
        fx = (fx / a.width) * a.scale;
        fy = (fy / a.height) * a.scale;
      }
      const float ex = np_clip((np_clip(fx, -a.clamp, a.clamp) + a.clamp) / a.two_clamp, 0.0f, 1.0f);
      const float ey = np_clip((np_clip(fy, -a.clamp, a.clamp) + a.clamp) / a.two_clamp, 0.0f, 1.0f);
      r = np_u8(ex * 255.0f);
      g = np_u8(ey * 255.0f);
      b = 0;
    }
    a.out[3 * p] = r;
    a.out[3 * p + 1] = g;
    a.out[3 * p + 2] = b;
  }
}
}  // namespace

extern "C" int vfml_flow_encode(const float* flow, int h, int w, int mode, float width, float height, float scale,
                                float clamp, float two_clamp, unsigned char* out, void* stream) {
  VFML_REQUIRE(flow && out && h > 0 && w > 0, "vfml_flow_encode: bad argument");
  VFML_REQUIRE(mode == VFML_ENCODE_GAMEDEV || mode == VFML_ENCODE_RG8 || mode == VFML_ENCODE_RGB8,
               "vfml_flow_encode: unknown mode %d", mode);
  EncodeArgs a;
  a.flow = flow; a.out = out; a.n = (int64_t)h * w; a.mode = mode;
  a.width = width; a.height = height; a.scale = scale; a.clamp = clamp; a.two_clamp = two_clamp;
  hipLaunchKernelGGL(flow_encode_kernel, dim3(grid_for(a.n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return vfml_check_launch("vfml_flow_encode");
}

extern "C" int vfml_coords_init(float* coords1, int n, int h, int w, void* stream) {
  VFML_REQUIRE(coords1 && n > 0 && h > 0 && w > 0, "vfml_coords_init: bad argument");
  VFML_REQUIRE(vfml_aligned16(coords1), "vfml_coords_init: alignment");
  const int64_t total = (int64_t)n * h * w;
  hipLaunchKernelGGL(coords_init_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (f32x4*)coords1, h, w, total);
  return vfml_check_launch("vfml_coords_init");
}

extern "C" int vfml_coords_update(float* coords1, const float* delta, int n, int h, int w, float* flow_a, int ld_a,
                                  float* flow_b, int ld_b, int fmt_b, void* stream) {
  VFML_REQUIRE(fmt_b == VFML_FMT_F32 || fmt_b == VFML_FMT_S16, "vfml_coords_update: bad fmt_b");
  if (fmt_b == VFML_FMT_S16 && flow_b)
    VFML_REQUIRE((reinterpret_cast<uintptr_t>(flow_b) & 31u) == 16 && ld_b % 8 == 0,
                 "vfml_coords_update: split-row flow_b must point at channel 4 of a unit (ld_b %% 8 == 0)");
  VFML_REQUIRE(coords1 && n > 0 && h > 0 && w > 0, "vfml_coords_update: bad argument");
  VFML_REQUIRE(vfml_aligned16(coords1) && vfml_aligned16(delta) && vfml_aligned16(flow_a) && vfml_aligned16(flow_b),
               "vfml_coords_update: alignment");
  VFML_REQUIRE((!flow_a || (ld_a >= 4 && ld_a % 4 == 0)) && (!flow_b || (ld_b >= 4 && ld_b % 4 == 0)),
               "vfml_coords_update: ld must be a multiple of 4 and >= 4");
  const int64_t total = (int64_t)n * h * w;
  hipLaunchKernelGGL(coords_update_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (f32x4*)coords1, (const f32x4*)delta, h, w, total, flow_a,
                     ld_a, flow_b, ld_b, fmt_b == VFML_FMT_S16 ? 1 : 0);
  return vfml_check_launch("vfml_coords_update");
}

extern "C" int vfml_flow_rows7(const float* flow, int n, int h, int w, float* rows, void* stream) {
  VFML_REQUIRE(flow && rows && n > 0 && h > 0 && w > 0, "vfml_flow_rows7: bad argument");
  VFML_REQUIRE(vfml_aligned16(flow) && (reinterpret_cast<uintptr_t>(rows) & 31u) == 0, "vfml_flow_rows7: flow 16-byte, rows 32-byte aligned");
  const int64_t total = (int64_t)n * h * w;
  hipLaunchKernelGGL(flow_rows7_kernel, dim3(grid_for(total * 4, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const f32x4*)flow, w, total, reinterpret_cast<char*>(rows));
  return vfml_check_launch("vfml_flow_rows7");
}

extern "C" int vfml_tapsum3x3(const float* t, int ld_t, const float* bias, int n, int h, int w, float* out, int parts,
                              int64_t part_stride, void* stream) {
  VFML_REQUIRE(t && out && n > 0 && h > 0 && w > 0, "vfml_tapsum3x3: bad argument");
  VFML_REQUIRE(parts >= 1 && parts <= 4 && (parts == 1 || (part_stride > 0 && part_stride % 4 == 0)),
               "vfml_tapsum3x3: parts in 1..4, part_stride a positive multiple of 4 floats");
  VFML_REQUIRE(ld_t >= 36 && ld_t % 4 == 0 && vfml_aligned16(t) && vfml_aligned16(out),
               "vfml_tapsum3x3: ld_t must be a multiple of 4 and >= 36, t and out 16-byte aligned");
  const int64_t total = (int64_t)n * h * w;
  hipLaunchKernelGGL(tapsum3x3_kernel, dim3(grid_for(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t,
                     ld_t, bias, h, w, total, (f32x4*)out, parts, part_stride, (f32x4*)nullptr, (float*)nullptr, 0, (float*)nullptr,
                     0, 0);
  return vfml_check_launch("vfml_tapsum3x3");
}

extern "C" int vfml_tapsum3x3_update(const float* t, int ld_t, const float* bias, int n, int h, int w, int parts,
                                     int64_t part_stride, float* coords1, float* flow_a, int ld_a, float* flow_b, int ld_b,
                                     int fmt_b, void* stream) {
  VFML_REQUIRE(t && coords1 && n > 0 && h > 0 && w > 0, "vfml_tapsum3x3_update: bad argument");
  VFML_REQUIRE(parts >= 1 && parts <= 4 && (parts == 1 || (part_stride > 0 && part_stride % 4 == 0)),
               "vfml_tapsum3x3_update: parts in 1..4, part_stride a positive multiple of 4 floats");
  VFML_REQUIRE(ld_t >= 36 && ld_t % 4 == 0 && vfml_aligned16(t), "vfml_tapsum3x3_update: ld_t must be a multiple of 4 and >= 36, t 16-byte aligned");
  VFML_REQUIRE(fmt_b == VFML_FMT_F32 || fmt_b == VFML_FMT_S16, "vfml_tapsum3x3_update: bad fmt_b");
  if (fmt_b == VFML_FMT_S16 && flow_b)
    VFML_REQUIRE((reinterpret_cast<uintptr_t>(flow_b) & 31u) == 16 && ld_b % 8 == 0,
                 "vfml_tapsum3x3_update: split-row flow_b must point at channel 4 of a unit (ld_b %% 8 == 0)");
  VFML_REQUIRE(vfml_aligned16(coords1) && vfml_aligned16(flow_a) && vfml_aligned16(flow_b), "vfml_tapsum3x3_update: alignment");
  VFML_REQUIRE((!flow_a || (ld_a >= 4 && ld_a % 4 == 0)) && (!flow_b || (ld_b >= 4 && ld_b % 4 == 0)),
               "vfml_tapsum3x3_update: ld must be a multiple of 4 and >= 4");
  const int64_t total = (int64_t)n * h * w;
  hipLaunchKernelGGL(tapsum3x3_kernel, dim3(grid_for(total, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t,
                     ld_t, bias, h, w, total, (f32x4*)nullptr, parts, part_stride, (f32x4*)coords1, flow_a, ld_a, flow_b, ld_b,
                     fmt_b == VFML_FMT_S16 ? 1 : 0);
  return vfml_check_launch("vfml_tapsum3x3_update");
}

extern "C" int vfml_flow_lod(const float* flow, int h, int w, float* out, void* stream) {
  VFML_REQUIRE(flow && out && h > 0 && w > 0, "vfml_flow_lod: bad argument");
  VFML_REQUIRE((reinterpret_cast<uintptr_t>(flow) & 7u) == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0,
               "vfml_flow_lod: 8-byte alignment");
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  hipLaunchKernelGGL(flow_lod_kernel, dim3(grid_for((int64_t)ho * wo, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const float2*)flow, h, w, ho, wo, (float2*)out);
  return vfml_check_launch("vfml_flow_lod");
}

extern "C" int vfml_convex_upsample(const float* coords1, int ch, const float* mask, int ld_mask, int h, int w,
                                    float* out, void* stream) {
  VFML_REQUIRE(coords1 && mask && out, "vfml_convex_upsample: null pointer");
  VFML_REQUIRE((ch == 0 || ch == 2) && h > 0 && w > 0 && ld_mask >= 576, "vfml_convex_upsample: bad ch/h/w/ld_mask");
  VFML_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0, "vfml_convex_upsample: out must be 8-byte aligned");
  hipLaunchKernelGGL(convex_upsample_kernel, dim3((h * w + 3) / 4), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), coords1, ch, mask, ld_mask, h, w, out);
  return vfml_check_launch("vfml_convex_upsample");
}
