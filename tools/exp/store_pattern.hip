// Dev experiment: HBM write bandwidth of tiled store patterns (who writes which 16 bytes when).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out [rows][ld] f32; tiles of tr x tc floats; 256 threads per workgroup, persistent grid.
// order 0: tile id n-fastest over the whole grid (concurrent workgroups = consecutive tiles)
// order 1: XCD x owns a contiguous chunk of the tile space (as the conv kernel does)
// order 2: m-fastest (concurrent workgroups = consecutive row tiles of one column tile)
__global__ __launch_bounds__(256) void store_pattern(float* out, int rows, int cols, int64_t ld, int tr, int tc, int order, int nt_hint) {
  const int mt = (rows + tr - 1) / tr, ntl = (cols + tc - 1) / tc;
  const int total = mt * ntl;
  const int G = gridDim.x;
  int tile, end, step;
  if (order == 1) {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int q = total >> 3, r = total & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    step = (G - xcd + 7) >> 3; tile = start + lw; end = start + q + (xcd < r ? 1 : 0);
  } else { tile = blockIdx.x; end = total; step = G; }
  const int l4 = tc / 4;            // threads per tile row (tc <= 1024: checked by the launcher)
  const int rpp = 256 / l4;         // rows per pass
  const int c4 = (threadIdx.x % l4) * 4, rr = threadIdx.x / l4;
  for (; tile < end; tile += step) {
    int tm, tn;
    if (order == 2) { tm = tile % mt; tn = tile / mt; } else { tn = tile % ntl; tm = tile / ntl; }
    const int col = tn * tc + c4;
    if (col >= cols) continue;
    for (int r = rr; r < tr; r += rpp) {
      const int row = tm * tr + r;
      if (row >= rows) break;
      f32x4 v = {1.f, 2.f, 3.f, (float)tile};
      f32x4* p = reinterpret_cast<f32x4*>(out + (int64_t)row * ld + col);
      if (nt_hint) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}

extern "C" int run_store_pattern(float* out, int rows, int cols, int64_t ld, int tr, int tc, int order, int nt_hint, int grid, void* stream) {
  if (tc < 4 || tc > 1024 || tc % 4 || 256 % (tc / 4)) return -1;   // a tile row is written by 1..256 threads
  hipLaunchKernelGGL(store_pattern, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, rows, cols, ld, tr, tc, order, nt_hint);
  return (int)hipGetLastError();
}
