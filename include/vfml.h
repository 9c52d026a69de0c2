/* vfml.h — C ABI of libvfml_hip.so, the MI355X (gfx950) kernels behind the multi-frame
 * optical-flow hot path.
 *
 * The reference (IvanPopov/video-flow-ml) has no FFI of its own: its boundary is the Python
 * object protocol `build_network(cfg)(images, {})` (processing/videoflow_core.py:28,101,188).
 * Everything below that call is PyTorch-CUDA library work (cuDNN conv, cuBLAS bmm, grid_sample,
 * unfold, softmax) launched by the un-vendored VideoFlow submodule.  Each entry point here
 * replaces one of those implicit library ops; the Python host
 * (video-flow-ml_amd/vfml/hip.py) binds them with ctypes and sequences them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA tensor .data_ptr()) unless
 *     the name says host; buffers are borrowed for the duration of the call only;
 *   - activations are NHWC fp32: tensor[n][y][x][c], `ld` = floats between consecutive pixels
 *     (>= channels; lets a kernel read or write a channel slice of a wider buffer);
 *   - channel counts, ld values and base pointers are multiples of 4 floats / 16 bytes;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - return 0 on success, non-zero on a rejected argument or launch error
 *     (text from vfml_last_error(), thread-local).  Nothing is launched on error.
 */
#ifndef VFML_H
#define VFML_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFML_ABI_VERSION 25

/* Epilogue selector of vfml_conv2d.  v = out_scale * (acc + addend[p][c] + bias[c]). */
enum {
  VFML_EPI_NONE = 0,       /* out = v                                                        */
  VFML_EPI_RELU = 1,       /* out = max(v, 0)                                                */
  VFML_EPI_TANH = 2,
  VFML_EPI_SIGMOID = 3,
  VFML_EPI_TANH_RELU = 4,  /* c <  split: tanh(v)   else relu(v)   (cnet -> net | inp)       */
  VFML_EPI_GRU_ZR = 5,     /* c <  split: sigmoid(v) else sigmoid(v) * aux0[p][c - split]    */
  VFML_EPI_GRU_Q = 6,      /* out = (1 - z) * h + z * tanh(v), z = aux0[p][c], h = aux1[p][c] */
  VFML_EPI_ADD_AUX = 7,    /* out = aux0[p][c] + v   (residual / memory read-out: m + gamma*acc)   */
};

/* Activation storage formats.
 *   VFML_FMT_F32  plain NHWC float32.
 *   VFML_FMT_S16  "split rows": same addressing and the same 4 bytes per channel, but every group of
 *                 8 channels (32 bytes, a "unit") holds 8 f16 hi halves then 8 f16 lo halves with
 *                 x = hi + lo (hi = f16(x) rounded to nearest, lo = f16(x - hi)): the operand format of the
 *                 split-f16 MFMA kernel, written once by the producer instead of being re-derived by
 *                 every consumer.  Channel offsets / counts / ld are multiples of 8 (a producer may
 *                 write a 4-channel half unit), bases 32-byte aligned. */
enum { VFML_FMT_F32 = 0, VFML_FMT_S16 = 1,
/* one f16 (round to nearest) per element, row pitches counted in ELEMENTS: only as the output of vfml_conv2d_split's GEMM
 * form (out and out_t: the correlation volumes) and as `vol_fmt` of the lookups that read them back - the opt-in half-size
 * correlation pyramid (cfg.corr_volume = "f16": half the store bytes of the volume GEMMs, half the pyramid memory, fewer
 * sectors per gathered window; the volume's values then carry 11 bits) */
       VFML_FMT_F16 = 2 };

/* Implicit-GEMM 2-D convolution / plain GEMM on the f32 matrix cores.
 *   out[p][co] = epi( sum_{ky,kx,ci} in(p; ky,kx)[ci] * w[co][ky][kx][ci] + bias[co] )
 * The input is the channel-concatenation of up to two NHWC sources (in1 may be NULL).
 * A 1x1/stride-1 conv with N*H*W = rows is a row-major GEMM  out[rows][cout] = A[rows][K]·W[cout][K]^T
 * (used for the all-pairs correlation volume).
 * Replaces: cuDNN conv2d + bias + activation, cuBLAS bmm. */
typedef struct vfml_conv_desc {
  const float* in0; int32_t c0; int32_t ld0;
  const float* in1; int32_t c1; int32_t ld1;       /* optional second source (NULL, 0, 0)     */
  int32_t n, h, w;                                  /* input batch / height / width            */
  const float* weight;                              /* [cout][kh][kw][c0+c1]                   */
  const float* bias;                                /* [cout] or NULL                          */
  int32_t cout, kh, kw, stride, pad_h, pad_w;
  float* out; int32_t ldo;                          /* [n][ho][wo][ldo], ho=(h+2ph-kh)/s+1     */
  int32_t epilogue; int32_t split; float out_scale;
  const float* aux0; int32_t ld_aux0;
  const float* aux1; int32_t ld_aux1;
  const float* addend; int32_t ld_addend;           /* optional f32 map [pixels][ld_addend] added to the
                                                       accumulator before bias/epilogue (a per-pixel bias:
                                                       the part of a convolution whose input does not
                                                       change between calls, computed once)             */
  float* out_t; int32_t ld_out_t;                   /* optional (vfml_conv2d_split, GEMM form only): the
                                                       TRANSPOSE of the result, out_t[co][pixel], row stride
                                                       ld_out_t floats - the all-pairs correlation of frame
                                                       pair (a, b) read the other way round is the volume of
                                                       (b, a): one pass of MFMAs, two stores             */
  int32_t flags;                                    /* VFML_CONV_* bits (vfml_conv2d_split)               */
  double* stats_part;                               /* optional (vfml_conv2d_split, plain f32 output): per block of G
                                                       consecutive output pixels and output channel, the sum and
                                                       the sum of squares of the stored result,
                                                       [n][ceil(hw/G)][cout][2] doubles (hw = output pixels per
                                                       image; n == 1 or hw % G == 0; G = 128 with f32 sources,
                                                       32 with split-row sources - VFML_STATS_ROWS_*): the first
                                                       pass of vfml_instnorm_stats done where the tile still is
                                                       in LDS; fold with vfml_instnorm_finalize               */
  float* ksplit_ws;                                 /* optional (vfml_conv2d_split, GEMM form, plain f32 out): a workspace
                                                       shaped like out ([n*h*w][ldo] floats), followed - when out_t is
                                                       given - by one shaped like out_t ([cout][ld_out_t]).  A GEMM with
                                                       few output tiles and a long K axis (the MemFlow read-out: 254
                                                       tiles on 512 resident slots, K = 32 400) is then run as two work
                                                       items per tile, one per half of K: the second half's sums go to
                                                       the workspace and are added to out / out_t (first + second, a
                                                       fixed order) by passes the call launches itself */
  /* optional projection epilogue (vfml_conv2d_split; split-row sources, VFML_EPI_RELU, no addend, cout % 128 == 0, the
     full split product): relu(out) is NOT stored (out is not written).  Per 128-column tile t of the output the kernel
     multiplies the tile, still in LDS, by that tile's slice of a second weight [proj_n][cout] and stores the partial sums
         proj_out[t][p][j] = sum_{c in 128 t .. 128 t + 127} relu(out[p][c]) * proj_w[j][c],   j < proj_n <= 48,
     proj_out = [cout / 128][n*ho*wo][ld_proj] floats; the caller adds the cout / 128 maps (vfml_tapsum3x3 does).  proj_hi /
     proj_lo: the f16 planes [proj_n][proj_kp] of proj_w * proj_scale (vfml_split_f16; lo behind hi within 1 GiB), three
     MFMAs per product (with VFML_CONV_MFMA2A two, in the projection as in the convolution: relu(out) as plain f16).  The flow head - 3x3 to 256 channels, ReLU, 256 -> 4 over 3x3 as a 1x1 to 36 tap-major columns -
     as ONE launch whose 256-channel map never travels to HBM and back.  Replaces: the second cuDNN conv of
     FlowHead. */
  const void* proj_hi; const void* proj_lo; int32_t proj_n; int32_t proj_kp; float proj_scale;
  float* proj_out; int32_t ld_proj;
  /* optional (vfml_conv2d_split): a DEVICE cell (8-byte aligned) holding the addend pointer to use, read when the kernel
     runs - the launch can sit in a replayed HIP graph while the per-pixel bias it adds (the context part of a GRU gate
     convolution, kept per frame) lives somewhere else from field to field.  `addend` must still be given: a pointer of the
     same shape and alignment, which is what the call validates (vfml_ptr_table_set writes such cells). */
  const float* const* addend_ind;
} vfml_conv_desc;

/* flags: accumulate the two cross terms of the split product in the order (a_lo*b_hi, a_hi*b_lo) instead of
 * (a_hi*b_lo, a_lo*b_hi).  out[q][s] of a call with operands (A, B) and out[s][q] of the call with operands
 * (B, A) and this flag are then the same sequence of f32 additions, i.e. bit-identical - which makes a
 * correlation volume computed directly equal to the one obtained as another call's out_t. */
enum { VFML_STATS_ROWS_F32 = 128, VFML_STATS_ROWS_S16 = 32 };
enum { VFML_CONV_SWAP_CROSS = 1,
/* Fewer MFMAs per product, per call (the per-layer precision plan of the network, cfg.precision):
 *   VFML_CONV_MFMA2  the weight operand as ONE f16 (its hi plane; hi is the round-to-nearest f16 of the weight, so the
 *                    dropped a*w_lo term is an unbiased 2^-12 relative perturbation of each weight): a_hi w + a_lo w;
 *   VFML_CONV_MFMA2A the ACTIVATION operand as one f16 instead: a_hi w_hi + a_hi w_lo.  The weights keep their 22 bits;
 *                    the activations' rounding is independent from pixel to pixel and iteration to iteration, where a
 *                    rounded weight is the same perturbation everywhere (measured at 1080p: 10-30x less end-point
 *                    error than VFML_CONV_MFMA2 on the same layer), and the activation operand is the larger stream;
 *   VFML_CONV_MFMA1  both operands as one f16: a_hi w_hi - plain f16 inputs, f32 accumulate ("fp16" arithmetic).
 * The lo halves that are not used are not fetched.  Where a tile shape / loader combination is not built for the
 * reduced count the call runs with three MFMAs (never less accurate than asked). */
       VFML_CONV_MFMA2 = 2, VFML_CONV_MFMA1 = 4, VFML_CONV_MFMA2A = 8,
/* Stage the activations once per TAP (conv_gemm_dma_kernel) even where the kernel that stages them once per filter
 * ROW and shifts the fragment reads (conv_gemm_tapx_kernel: stride-1 "same" convolutions over split-row sources with
 * 2..5 taps per row) would take the call.  Same arithmetic in the same order - bit-identical results; for A/B
 * measurements and the test that says so. */
       VFML_CONV_PER_TAP = 16 };

int vfml_conv2d(const vfml_conv_desc* d, void* stream);

/* Same contract on the f16 matrix cores with fp32-grade accuracy ("split-f16": x = hi + lo with
 * hi = f16(x), lo = f16(x - hi); a*b ~= ah*bh + ah*bl + al*bh, 3 MFMAs per product, ~22 mantissa
 * bits).  d->weight is ignored; w_hi / w_lo are the two f16 planes [cout][kp] made by
 * vfml_split_f16 from the [cout][K] weight matrix times `w_scale`, kp = K rounded up to a multiple
 * of 32; the kernel divides the accumulator by w_scale.  Activations stay fp32 in HBM and are split
 * while staged into LDS.
 * w_lo == NULL: the second operand is ONE plain f16 plane (up to 2 GiB), a product is two MFMAs (a_hi b + a_lo b);
 * GEMM form only (split-row source, 1x1 over whole 32-channel blocks, plain f32 out, cout >= 1024). */
int vfml_conv2d_split(const vfml_conv_desc* d, const void* w_hi, const void* w_lo, int kp, float w_scale,
                      int in_fmt, int out_fmt, int aux_fmt, int k_order, void* stream);
/* in_fmt: format of in0/in1; out_fmt: of out; aux_fmt: of aux0/aux1 (VFML_FMT_*).
 * k_order: order of the K axis of the weight planes.
 *   VFML_KORDER_TAP     [cout][ky][kx][ci], kp = kh*kw*(c0+c1) rounded up to 32 (as d->weight).
 *   VFML_KORDER_CBLOCK  [cout][ci/32][ky][kx][ci%32], channels zero padded to a multiple of 32,
 *                       kp = kh*kw*roundup32(c0+c1); split-row sources only.  All taps of one
 *                       32-channel block are consecutive K steps, so the kh*kw reads of an input line
 *                       happen while it is still in L2 (tap-major order re-reads it ctot/32 steps later,
 *                       from the Infinity Cache or HBM).  For 1x1 convolutions the two orders coincide.
 * Split-row sources (in_fmt VFML_FMT_S16) are staged by LDS-DMA: c0, c1, ld0, ld1 multiples of 8, at least 32
 * channels, and w_hi / w_lo within 1 GiB of each other (halves of one allocation, as vfml_split_f16's callers
 * make them) because both planes are read through one buffer descriptor.  A 1x1 / stride-1 convolution over ONE
 * split-row source (GEMM rows) may span any number of bytes (the descriptor is rebased per tile); every other
 * source must span < 1 GiB, both sources of a two-source call must lie within 2 GiB of each other.  Plain f32
 * outputs at least 1024 channels wide (cout % 4 == 0) take the persistent GEMM form of the kernel. */
/*   VFML_KORDER_CBLOCK64 [cout][ci/64][ky][kx][ci%64]: the order of VFML_CONV_MFMA1 calls whose sources are whole 64-channel
 *                       blocks (c0, c0+c1 multiples of 64; kp = K): the kernel then steps 64 channels of hi halves at a
 *                       time (half the K steps and LDS-DMA instructions of the 32-channel steps, no lo half fetched). */
enum { VFML_KORDER_TAP = 0, VFML_KORDER_CBLOCK = 1, VFML_KORDER_CBLOCK64 = 2 };

/* scale * f32 rows [rows][c] (row stride ld_src floats) -> split rows [rows][ld_dst] (VFML_FMT_S16); c % 4 == 0.
 * (scale: a power of two keeps the split exact - the correlation GEMM's query rows carry x16.) */
int vfml_to_s16(const float* src, int64_t rows, int c, int ld_src, float* dst, int ld_dst, float scale, void* stream);

/* y = scale * src (f32 [rows][k], row stride ld floats) -> hi = f16(y), lo = f16(y - hi), each
 * [rows][kp], zero padded from k to kp (kp % 32 == 0).  A power-of-two scale that brings max|y|
 * near 2^14 keeps the lo halves out of the f16 subnormal range. */
int vfml_split_f16(const float* src, int64_t rows, int k, int ld, float scale, void* hi, void* lo, int kp,
                   void* stream);

/* Row softmax: out[r][c] = scale * exp(x[r][c] - max_r) / sum_r for c < cols, written as split rows
 * (VFML_FMT_S16) and zero-filled up to ld_out (ld_out % 8 == 0, ld_out >= cols).  x: f32 [rows][ld_in].
 * scale (a power of two <= 2^15; the consumer divides it out through out_scale) keeps small probabilities out of
 * the f16 subnormal range: a 1080p row has 32400 of them, 3e-5 on average, and unscaled every one would lose up to
 * 6e-8 - 0.1 % of the row's mass.
 * Replaces: torch.softmax over the key axis of the attention scores (materialised:
 * with 288 GB of HBM the P x P score matrix of one frame simply stays resident). */
int vfml_softmax_rows_s16(const float* x, int64_t rows, int cols, int64_t ld_in, float* out, int64_t ld_out, float scale,
                          void* stream);

/* The same softmax as ONE round-to-nearest f16 per element, rows of ld_out halves (ld_out % 8 == 0, <= 32768): the
 * hi-only weight plane of vfml_conv2d_split (w_lo == NULL).  For MemFlow's attention matrix this halves the bytes
 * its 12 read-outs per field stream; 2^-12 relative per probability, unbiased (measured: 1080p EPE unchanged). */
int vfml_softmax_rows_f16(const float* x, int64_t rows, int cols, int64_t ld_in, void* out, int64_t ld_out, float scale,
                          void* stream);

/* src f32 [rows][c] (row stride ld) -> SPLIT ROWS of its transpose times scale: dst[c][ld_dst] (VFML_FMT_S16, ld_dst =
 * rows rounded up to 32, pad channels zero): the activation operand of out^T = V^T . A^T. */
int vfml_transpose_to_s16(const float* src, int rows, int c, int ld, float scale, float* dst, int64_t ld_dst, void* stream);

/* out = aux + scale * x:  x f32 [rows][ldx], aux / out split rows (c % 8 == 0 channels). */
int vfml_add_to_s16(const float* x, int64_t ldx, const float* aux, int64_t ld_aux, float* out, int64_t ld_out, int64_t rows,
                    int c, float scale, void* stream);

/* src f32 [rows][c] (row stride ld) -> split-f16 planes of its TRANSPOSE times scale: hi/lo [c][kp],
 * kp >= rows, kp % 32 == 0, zero padded: the "weight" operand of out = attn . V. */
int vfml_transpose_split_f16(const float* src, int rows, int c, int ld, float scale, void* hi, void* lo, int kp,
                             void* stream);

/* K1: frames -> normalised NHWC4 (4th channel zero):  dst = scale * x + shift.
 * kind 0: src is uint8 [n][H][W][3] (values 0..255, x = u8/255 as the reference does at
 *         processing/videoflow_processor.py:154);  kind 1: src is float32 [n][3][H][W]. */
int vfml_frames_to_nhwc4(const void* src, int kind, int n, int H, int W,
                         float scale, float shift, float* dst, void* stream);

/* Instance norm (affine-free, eps, biased variance) over NHWC x[n][hw][c] (dense, ld == c).
 * stats[n][c] = {mean, rstd}; workspace >= vfml_instnorm_workspace_bytes(n, hw, c).
 * Replaces: nn.InstanceNorm2d in the encoders. */
int64_t vfml_instnorm_workspace_bytes(int n, int hw, int c);
int vfml_instnorm_stats(const float* x, int n, int hw, int c, float eps,
                        float* stats, void* workspace, void* stream);
/* out = relu( norm(x; stats) )                                  if res == NULL
 * out = relu( res' + relu(norm(x; stats)) )                     otherwise, where
 *       res' = res (res_stats == NULL) or norm(res; res_stats)  (down-sampled shortcut).
 * out_fmt VFML_FMT_S16: out - and a res without res_stats, which is an earlier out - are split rows (c % 8 == 0): the
 * operand format of the LDS-DMA convolution kernel that consumes them; x (a convolution's raw result) and a res with
 * res_stats stay f32. */
int vfml_instnorm_apply(const float* x, const float* stats, const float* res,
                        const float* res_stats, int n, int hw, int c, float* out, int out_fmt, void* stream);

/* 2x2/stride-2 average pooling (floor) of an NHWC map; used to build the pooled target-feature
 * pyramid so that pyramid level l of the correlation volume is one GEMM against level-l
 * features (avg-pool commutes with the dot product). */
int vfml_avgpool2x2(const float* x, int n, int h, int w, int c, float* out, void* stream);

/* K5: correlation lookup.  nmaps query maps (correlation problems) of q_per_map queries each; query q
 * of map m reads row q of that map's pyramid.  For each query sample a (2r+1)^2 window around
 * coords/2^l with bilinear interpolation, zeros outside, RAFT's window order (channel i*(2r+1)+j
 * samples x+d[i], y+d[j]).
 *   pyr[m*levels+l] : float [q_per_map][ld[l]]  (columns = hl[l]*wl[l] targets, row-major); host array
 *                     of device pointers - each problem's pyramid may live in its own allocation
 *   vol_tile        : 0, or tws + 16 * ths: every level image is stored as tiles of 2^tws x 2^ths texels (tile after
 *                     tile, left to right then down; a tile row-major; edge tiles whole, so ld[l] >= the tiles' texels),
 *                     and so is the query grid that orders the ROWS: query q = (y, x) of the hl[0] x wl[0] grid reads
 *                     row tile_position(y, x) (q_per_map == hl[0]*wl[0]; rows of pad positions are never read).  What the
 *                     volume GEMM writes when both its operands' rows are in that order (the engine: 4 x 8 tiles - a
 *                     lookup window then lies in ~8 lines of 128 bytes instead of ~13)
 *   coords          : float [nmaps*q_per_map][ld_coords], x at +0, y at +1
 *   out             : [nmaps*q_per_map][ld_out], levels*(2r+1)^2 channels written from out
 * Replaces: F.grid_sample(align_corners=True) x levels. */
int vfml_corr_lookup(const float* const* pyr, const int32_t* hl, const int32_t* wl,
                     const int32_t* ld, int levels, int radius, int nmaps, int q_per_map,
                     const float* coords, int ld_coords, float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile,
                     void* stream);
/* out_fmt VFML_FMT_S16: channels are written as split rows; the channel count is rounded up to a
 * multiple of 8 with zero channels (out is 32-byte aligned, ld_out % 8 == 0).
 * vol_fmt VFML_FMT_F32, or VFML_FMT_F16: the pyramids hold one f16 per element (ld in elements; radius 3 or 4, at most four
 * levels), or VFML_VOL_F16_LEVELS(m): only the levels whose bit is set in m do (m = 14: levels 1-3, 12: levels 2-3, 8: level
 * 3; the coarse levels cost the fewest bits of the flow and the lookup reads as many texels of each level as of level 0). */
#define VFML_VOL_F16_LEVELS(m) (0x100 | (m))

/* The same lookup with the pyramid pointers read from a DEVICE table at run time: table[m * levels + l] is what
 * pyr[m * levels + l] is above.  A launch recorded in a HIP graph (the update iterations of a field are a fixed launch
 * sequence, replayed per field) then follows whatever pyramids the table names when the graph runs.
 * vfml_ptr_table_set writes n <= 48 device pointers into such a table, asynchronously on `stream` (the values travel as
 * kernel arguments: no host buffer has to outlive the call). */
int vfml_corr_lookup_indirect(const float* const* table, const int32_t* hl, const int32_t* wl, const int32_t* ld,
                              int levels, int radius, int nmaps, int q_per_map, const float* coords, int ld_coords,
                              float* out, int ld_out, int out_fmt, int vol_fmt, int vol_tile, void* stream);
/* Both directions of a window's centre frames in ONE launch: `table` holds the forward problems' pyramids of the query maps
 * from entry 0 and their backward problems' from entry dir_tab * levels (dir_tab >= nmaps; 2 * nmaps <= 8 maps per launch).
 * The second direction of query map m reads the coordinates at coords[row * ld_coords + dir_coords ..] and writes to
 * out[row * ld_out + dir_out ..] - what two calls of vfml_corr_lookup_indirect with shifted table / coords / out pointers
 * do, bit for bit. */
int vfml_corr_lookup_indirect_bidir(const float* const* table, const int32_t* hl, const int32_t* wl, const int32_t* ld,
                                    int levels, int radius, int nmaps, int q_per_map, const float* coords, int ld_coords,
                                    int dir_coords, float* out, int ld_out, int dir_out, int dir_tab, int out_fmt, int vol_fmt,
                                    int vol_tile, void* stream);
int vfml_ptr_table_set(void* table, const void* const* ptrs, int n, void* stream);

/* The 7x7 convolution over the 4-channel flow map (motion encoder, convf1) as a 7x1 convolution over 32 channels: this pass
 * writes, per pixel, the seven horizontal taps' flow quads (zeros outside the image row) as one split-row block of 32
 * channels - channel kx*4 + c = flow[y][x + kx - 3][c], channels 28..31 zero - 12 MB at 1080p/8; the vertical taps are then
 * the LDS-DMA convolution kernel's own (kh = 7, kw = 1, weights repacked to [cout][ky][kx*4 + c]).  flow: [n*h*w][4] f32,
 * rows: [n*h*w][32] split rows (VFML_FMT_S16, 32-byte aligned). */
int vfml_flow_rows7(const float* flow, int n, int h, int w, float* rows, void* stream);

/* The encoders' 64 -> 64 channel 3x3 convolution (stride 1, padding 1; K2: the residual blocks of `layer1`) with persistent
 * workgroups that keep the weights in registers and one input patch per 4 x 32-pixel tile in LDS:
 *   out[p][0..63] = conv3x3(in)[p] + bias   (plain f32 rows, ldo floats apart),
 * in: split rows (VFML_FMT_S16), the 64 channels from in[p * ld_in]; w_hi / w_lo: the f16 planes [64][576] of the weights *
 * w_scale in VFML_KORDER_CBLOCK order; stats_part (optional): as vfml_conv_desc.stats_part with VFML_STATS_ROWS_S16 (one
 * {sum, sum of squares} pair per 32 consecutive pixels and channel).  The image width must be a multiple of 32.  Full split
 * product, the K order of vfml_conv2d_split: the result and the partial sums are bit-identical to that call's. */
int vfml_conv3x3_c64(const float* in, int ld_in, int n, int h, int w, const void* w_hi, const void* w_lo, int kp, float w_scale,
                     const float* bias, float* out, int ldo, double* stats_part, void* stream);

/* The flow half of the motion encoder as one launch (K6, BasicMotionEncoder.convf1 + convf2 with their ReLUs), for plans that
 * run both layers with ONE MFMA per product (VFML_CONV_MFMA1: plain f16 operands, f32 accumulate):
 *   out[p][0..63] = relu(conv3x3(relu(conv7x7(flow) + b1)) + b2),   flow: [n*h*w][4] f32,
 * out: split rows (VFML_FMT_S16), 64 channels from out[p * ld_out] (32-byte aligned, ld_out % 8 == 0).  w1_hi: the f16 hi
 * plane [128][224] of convf1's weights * w1_scale in the vfml_flow_rows7 layout (K = ky * 32 + kx * 4 + c), w2_hi: the hi
 * plane [64][1152] of convf2's * w2_scale in VFML_KORDER_CBLOCK64 order (K = (c / 64) * 576 + (ky * 3 + kx) * 64 + c % 64);
 * kp1 / kp2 their row pitches in halves (224 / 1152).  The 128-channel map between the two layers lives in LDS only; the
 * result is bit-identical to vfml_flow_rows7 + two vfml_conv2d_split calls with VFML_CONV_MFMA1. */
int vfml_flow_half(const float* flow, int n, int h, int w, const void* w1_hi, int kp1, float w1_scale, const float* b1,
                   const void* w2_hi, int kp2, float w2_scale, const float* b2, float* out, int ld_out, void* stream);

/* A 3x3 "same" convolution with FOUR output channels as a 1x1 convolution to 36 (tap-major: column (ky*3+kx)*4 + o holds
 * sum_c w[o][c][ky][kx] x[.][c]) followed by this pass: out[p][o] = bias[o] + sum over the nine taps inside the image of
 * t[p + (ky-1) w + (kx-1)][(ky*3+kx)*4 + o], taps in ky-major order.  The update block's flow head (256 -> 4) costs nine
 * times fewer MFMA steps this way than as a 3x3 convolution padded to a 32-column tile.  t: [n*h*w][ld_t] f32 (ld_t >= 36,
 * multiple of 4, 16-byte aligned rows), bias: 4 floats or NULL, out: [n*h*w][4] f32.  parts (1..4): t is that many such maps
 * part_stride floats apart (a multiple of 4) whose sum is meant - the partial maps of vfml_conv_desc.proj_out; per tap they
 * are added in order, map 0 first. */
int vfml_tapsum3x3(const float* t, int ld_t, const float* bias, int n, int h, int w, float* out, int parts, int64_t part_stride,
                   void* stream);
/* The same sum used at once as an iteration's flow update: coords1[p] += it, then the flows are emitted as by
 * vfml_coords_update(coords1, delta = that sum, ...) - one launch instead of two, same values. */
int vfml_tapsum3x3_update(const float* t, int ld_t, const float* bias, int n, int h, int w, int parts, int64_t part_stride,
                          float* coords1, float* flow_a, int ld_a, float* flow_b, int ld_b, int fmt_b, void* stream);

/* coords1 += delta (4 floats per pixel: fwd x,y, bwd x,y); flow = coords1 - grid is written to
 * flow_a[p*ld_a..+4] and flow_b[p*ld_b..+4] (either may be NULL).  h,w give the pixel grid,
 * n maps. delta may be NULL (just (re)emit flow).  */
int vfml_coords_update(float* coords1, const float* delta, int n, int h, int w,
                       float* flow_a, int ld_a, float* flow_b, int ld_b, int fmt_b, void* stream);
/* fmt_b VFML_FMT_S16: flow_b points at channel 4 of a split-row unit (16-byte aligned + 8): the four
 * flow values become the second quad of that unit (hi at +0..7, lo at +16..23 of the quad slot). */
/* coords1[p] = (x, y, x, y) */
int vfml_coords_init(float* coords1, int n, int h, int w, void* stream);

/* K8: 8x convex upsampling of one flow (2 of the 4 coords channels) of one map.
 *   coords1: [h][w][4] of that map (flow = coords1[ch..ch+1] - grid), mask: [h][w][ld_mask]
 *   (576 logits used: tap k (3x3, row-major) * 64 + sy*8 + sx), out: [8h][8w][2] (HWC).
 * Replaces: softmax + F.unfold + sum + permute and the CHW->HWC permute of
 * processing/videoflow_processor.py:185. */
int vfml_convex_upsample(const float* coords1, int ch, const float* mask, int ld_mask,
                         int h, int w, float* out, void* stream);

/* Second pass of vfml_instnorm_stats on partial sums a convolution left behind (vfml_conv_desc.stats_part):
 * part [n][chunks][c][2] doubles -> stats [n][c][2] = {mean, 1/sqrt(var + eps)}, folded in a fixed order.
 * With many partials per channel (chunks >= 1024, c % 8 == 0) they are first folded slice-wise with coalesced reads into
 * `workspace` (caller-owned device memory, at least vfml_instnorm_finalize_workspace_bytes(chunks, c) bytes; a larger
 * one lets more frames of a batch go through per launch).  The route depends on (chunks, c) alone - never on n or on
 * the workspace's size - so a frame's statistics are the same bits alone or in a batch; where the fold route applies a
 * NULL / too small workspace is an error, not another route.  The library keeps no hidden device allocation: calls on
 * different streams are independent as long as their workspaces are (ABI 23; until ABI 22 a per-device static scratch). */
int64_t vfml_instnorm_finalize_workspace_bytes(int chunks, int c);   /* 0: this shape folds without a workspace */
int vfml_instnorm_finalize(const double* part, int n, int chunks, int c, int hw, float eps, float* stats,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* The encoders' stem (K2's first layer): 7x7 convolution, stride 2, padding 3, 4 -> 64 channels, over NHWC4 frames
 * (vfml_frames_to_nhwc4's output) in the split-f16 arithmetic (three MFMAs per product), as one kernel that keeps a tile's
 * input patch and all weights in LDS (csrc/stem.hip).
 *   frames      [n][h][w][4] f32;  out [n][ho][wo][64] f32, ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1
 *   w_hi, w_lo  f16 planes [64][224] of the weights times w_scale, K order ky-major: k = ky * 32 + kx * 4 + c with kx < 8
 *               (kx = 7: zeros) and c < 4 - what vfml_split_f16 makes of a [64][7][8][4] f32 tensor
 *   stats_part  NULL, or [n][vfml_stem7x7s2_chunks(h, w)][64][2] doubles: {sum, sum of squares} of the stored values per
 *               8 x 64-pixel output tile and channel - fold with vfml_instnorm_finalize(chunks = that count)
 * Replaces: the first nn.Conv2d(3, 64, 7, stride=2, padding=3) of the RAFT encoder. */
int vfml_stem7x7s2_chunks(int h, int w);
int vfml_stem7x7s2(const float* frames, int n, int h, int w, const void* w_hi, const void* w_lo, float w_scale,
                   const float* bias, float* out, double* stats_part, void* stream);

/* One level of the flow-cache LOD pyramid (reference storage/cache_manager.py:77-161): out[y][x] =
 * 0.5 * (sum of the 2x2 block's in-image vectors) / (number of in-image cells); odd sides are padded
 * bottom/right with weight 0.  flow: [h][w][2] f32, out: [(h+1)/2][(w+1)/2][2].  Same association as
 * the reference's per-block np.sum, divisions by 1, 2 or 4: bit-exact. */
int vfml_flow_lod(const float* flow, int h, int w, float* out, void* stream);

/* Flow field [h][w][2] f32 (pixels) -> 8-bit RGB image [h][w][3] as the reference's encoders produce it, byte for
 * byte (encoding/flow_encoders.py: GamedevFlowEncoder :70-117, MotionVectorsRG8FlowEncoder :120-153,
 * MotionVectorsRGB8FlowEncoder 'rgb+' :242-293).  gamedev: (flow / (width, height)) * scale, clamp, map to
 * [0,1]; rg8: clamp, map; rgb8: direction / magnitude code.  clamp / two_clamp are float32(clamp_range) and
 * float32(2 * clamp_range) (numpy promotes the Python floats that way).  SURVEY.md 8(f)-3. */
enum { VFML_ENCODE_GAMEDEV = 0, VFML_ENCODE_RG8 = 1, VFML_ENCODE_RGB8 = 2 };
int vfml_flow_encode(const float* flow, int h, int w, int mode, float width, float height, float scale,
                     float clamp, float two_clamp, unsigned char* out, void* stream);

/* One temporal-anti-aliasing step: blend the current frame into its history, the history reprojected along the flow
 * field (reference effects/taa_processor.py: apply_taa :42-89, _apply_flow_based_taa :92-146,
 * _bilateral_reprojection_sample :148-216, _bilinear_sample :218-263).  SURVEY.md 8(f)-3.
 *   current: [h][w][3] u8 or f32 (0..255); flow: [h][w][2] f32, pixels, pointing from a pixel to where it was in the
 *   history image (NULL in SIMPLE mode); history: [h][w][3] f32 or f64; out: [h][w][3], never the history buffer.
 *   SIMPLE    out = alpha*current + (1-alpha)*history                         (out type = history type)
 *   BILINEAR  history sampled bilinearly at (x, y) + flow, clamped to the image (out f32)
 *   BILATERAL the four neighbours weighted by bilinear weight x exp(-(lum - lum_k)^2 / (0.2 sigma_color^2 + 1e-6)),
 *             renormalised (out f64)
 * Every intermediate has the dtype numpy's promotion gives it in the reference, so a sequence of steps tracks the
 * reference's history to the last few ulps (exp() is the only step that is not reproduced bit for bit). */
enum { VFML_TAA_SIMPLE = 0, VFML_TAA_BILINEAR = 1, VFML_TAA_BILATERAL = 2 };
enum { VFML_PIX_U8 = 0, VFML_PIX_F32 = 1, VFML_PIX_F64 = 2 };
int vfml_taa_blend(const void* current, int cur_type, const float* flow, const void* history, int hist_type,
                   void* out, int out_type, int h, int w, int mode, double alpha, double sigma_color, void* stream);

/* Flow quality map (reference correction_worker.py: generate_quality_frame_gpu :175-208): how well frame2, warped
 * back along the flow, matches frame1.  frame1, frame2, out: [h][w][3] u8; flow: [fh][fw][2] f32 at the frame's
 * resolution or at a cached LOD's (then resized bilinearly, align_corners = False, and rescaled by w/fw, h/fh).
 * Per pixel: target = (x, y) - flow, truncated to a texel; similarity = mean of (1 - |d|_2 / 1.732),
 * (1 - mean |d|), (cos + 1) / 2 of the two colours in [0,1]; out = (0, 2 (s - 0.5), 0) * 255 if s > threshold else
 * ((1 - s), 0, 0) * 255; (255, 0, 0) where the target is outside the image.  SURVEY.md 8(f)-4. */
int vfml_flow_quality_map(const unsigned char* frame1, const unsigned char* frame2, const float* flow, int fh, int fw,
                          int h, int w, float threshold, unsigned char* out, void* stream);

const char* vfml_last_error(void);
int vfml_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
