// HBM-bound row kernels of the HMMC towers: LayerNorm forward/backward (K2), column sums
// (bias / positional-embedding gradients), patch extraction feeding the patch-embed GEMM (K1),
// CLS/positional add, token-embedding gather and its scatter-add backward.
// Reference: modules/module_clip.py:217-223 (LayerNorm, fp32 math on fp16 tensors),
// modules/until_module.py:54-67 (TF-style LN of the temporal blocks), modules/module_clip.py:307-313
// (conv1 patchify, class token, positional embedding), modules/module_cross.py:287-291 (token embedding).
// Every kernel moves 16 bytes per lane per access and keeps fp32 statistics in registers.
#include "common.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<half_t> { static constexpr int N = 8; typedef h8 type; };
template <> struct Vec<float> { static constexpr int N = 4; typedef f4 type; };

constexpr int LN_MAXD = 1024;  // widest LayerNorm row (ViT-B: 768, text/temporal: 512)

// ---- LayerNorm forward ------------------------------------------------------------------------------------------------
// A wave owns rows w, w + nwaves, ...: gamma and beta stay in registers for all of them, the next row is requested before
// the current one is reduced (two dependent wave reductions sit between a row's load and its store), and every load is
// unconditional - a lane whose chunk lies past the row's end re-reads chunk 0 and its values are masked out - because a
// branch around a load makes the compiler wait for every load in flight where the paths join.
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     const int* __restrict__ row_index, int rows, int D,
                                                     long in_stride, float eps) {
  constexpr int VN = Vec<T>::N;
  constexpr int LN_MAXV = LN_MAXD / 64 / VN;
  typedef typename Vec<T>::type V;
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int nchunk = D / VN;
  bool ok[LN_MAXV];
  int ce[LN_MAXV];
  float gm[LN_MAXV][VN], bt[LN_MAXV][VN];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    ok[i] = lane + 64 * i < nchunk;
    ce[i] = ok[i] ? (lane + 64 * i) * VN : 0;
#pragma unroll
    for (int j = 0; j < VN; ++j) { gm[i][j] = gamma[ce[i] + j]; bt[i][j] = beta[ce[i] + j]; }
  }
  auto load_row = [&](int row, V (&t)[LN_MAXV]) {
    const long src_row = row_index ? (long)row_index[row] : (long)row;
    const T* xr = x + src_row * in_stride;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) t[i] = *reinterpret_cast<const V*>(xr + ce[i]);
  };
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  V cur[LN_MAXV], nxt[LN_MAXV];
  load_row(row, cur);
  for (; row < rows; row += nwaves) {
    load_row(min(row + nwaves, rows - 1), nxt);          // past the last row: a harmless re-read
    float v[LN_MAXV][VN];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
      for (int j = 0; j < VN; ++j) { v[i][j] = (float)cur[i][j]; if (ok[i]) s += v[i][j]; }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
      for (int j = 0; j < VN; ++j) { float d = v[i][j] - mean; if (ok[i]) q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    T* yr = y + (long)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      V o;
#pragma unroll
      for (int j = 0; j < VN; ++j) o[j] = (T)((v[i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
      if (ok[i]) *reinterpret_cast<V*>(yr + ce[i]) = o;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) cur[i] = nxt[i];
  }
}

// ---- LayerNorm backward: dx (+ optional residual grad), per-block partial dgamma/dbeta ---------
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.
// Rows are read through the same optional row_index / in_stride as the forward; dx rows are
// written to dx + dst_row * out_stride where dst_row = row_index ? row_index[row] : row.
// Same structure as the forward: gamma in registers, the next row's x / dy / residual gradient requested before this row's
// two reductions, unconditional loads.
template <typename T>
__global__ __launch_bounds__(256, 3) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ partial,
                                                     const int* __restrict__ row_index, int rows, int D,
                                                     long in_stride, int want_dxsum) {
  constexpr int VN = Vec<T>::N;
  constexpr int LN_MAXV = LN_MAXD / 64 / VN;
  typedef typename Vec<T>::type V;
  __shared__ float sred[4 * LN_MAXD];   // one stripe per wave
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunk = D / VN;
  const int nwaves = gridDim.x * 4;
  bool ok[LN_MAXV];
  int ce[LN_MAXV];
  float gm[LN_MAXV][VN];
  float dg[LN_MAXV][VN], db[LN_MAXV][VN], ds[LN_MAXV][VN];   // ds: column sums of the dx rows written (bias gradient
#pragma unroll                                                //     of the linear layer that produced this activation)
  for (int i = 0; i < LN_MAXV; ++i) {
    ok[i] = lane + 64 * i < nchunk;
    ce[i] = ok[i] ? (lane + 64 * i) * VN : 0;
#pragma unroll
    for (int j = 0; j < VN; ++j) { gm[i][j] = gamma[ce[i] + j]; dg[i][j] = 0.f; db[i][j] = 0.f; ds[i][j] = 0.f; }
  }
  const bool has_res = dres != nullptr;
  auto load_row = [&](int row, V (&tx)[LN_MAXV], V (&td)[LN_MAXV], float& mean, float& rstd) {
    const long src_row = row_index ? (long)row_index[row] : (long)row;
    const T* xr = x + src_row * in_stride;
    const T* dyr = dy + (long)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      tx[i] = *reinterpret_cast<const V*>(xr + ce[i]);
      td[i] = *reinterpret_cast<const V*>(dyr + ce[i]);
    }
    mean = mean_in[row];
    rstd = rstd_in[row];
  };
  int row = blockIdx.x * 4 + w;
  V tx[LN_MAXV], td[LN_MAXV], tr[LN_MAXV], nx[LN_MAXV], nd[LN_MAXV];
  float mean = 0.f, rstd = 0.f, nmean, nrstd;
  if (row < rows) load_row(row, tx, td, mean, rstd);
  for (; row < rows; row += nwaves) {
    const long src_row = row_index ? (long)row_index[row] : (long)row;
    // this row's residual gradient (used after the two reductions below) and the next row's x / dy
    if (has_res) {
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) tr[i] = *reinterpret_cast<const V*>(dres + src_row * in_stride + ce[i]);
    }
    load_row(min(row + nwaves, rows - 1), nx, nd, nmean, nrstd);
    // the row stays in registers as loaded (packed); xhat and g are recomputed for the second sweep instead of
    // being held as fp32 arrays, which keeps 4 waves per SIMD resident
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < VN; ++j) {
          float d = (float)td[i][j];
          float xh = ((float)tx[i][j] - mean) * rstd;
          float g = d * gm[i][j];
          s1 += g;
          s2 += g * xh;
          dg[i][j] += d * xh;
          db[i][j] += d;
        }
      }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
    T* dxr = dx + src_row * in_stride;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      V o;
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        float xh = ((float)tx[i][j] - mean) * rstd;
        float g = (float)td[i][j] * gm[i][j];
        float val = rstd * (g - s1 - xh * s2);
        if (has_res) val += (float)tr[i][j];
        o[j] = (T)val;
        if (ok[i]) ds[i][j] += (float)o[j];
      }
      if (ok[i]) *reinterpret_cast<V*>(dxr + ce[i]) = o;
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) { tx[i] = nx[i]; td[i] = nd[i]; }
    mean = nmean; rstd = nrstd;
  }
  // combine the 4 waves of the block, then one partial row per block: partial[block][0|1|2][D]
  const int npass = want_dxsum ? 3 : 2;
  for (int pass = 0; pass < npass; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < VN; ++j) sred[w * LN_MAXD + ce[i] + j] = pass == 0 ? dg[i][j] : (pass == 1 ? db[i][j] : ds[i][j]);
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256) {
      float t = sred[col] + sred[LN_MAXD + col] + sred[2 * LN_MAXD + col] + sred[3 * LN_MAXD + col];
      partial[((long)blockIdx.x * npass + pass) * D + col] = t;
    }
  }
}

// out[col] = sum_r partial[r][col] for an fp32 [R][N] partial matrix.  32 columns x 8 row-lanes per block:
// each row-lane streams 128-byte row segments, the 8 lanes are combined through LDS (fixed order, so the
// result is deterministic).  Columns < split go to out0, the rest to out1 (LayerNorm: dgamma | dbeta).
template <typename TO>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ partial, TO* __restrict__ out0,
                                                        TO* __restrict__ out1, int R, int N, int split, int round_f16) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  float a0 = 0.f, a1 = 0.f;
  if (col < N) {
    int r = rl;
    for (; r + 8 < R; r += 16) {
      a0 += partial[(long)r * N + col];
      a1 += partial[(long)(r + 8) * N + col];
    }
    if (r < R) a0 += partial[(long)r * N + col];
  }
  red[rl][c] = a0 + a1;
  __syncthreads();
  if (rl == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    if (round_f16) t = r16(t);
    if (col < split) out0[col] = (TO)t;
    else out1[col - split] = (TO)t;
  }
}

// LayerNorm backward second stage: columns [0,D) -> dgamma, [D,2D) -> dbeta (fp32), [2D,3D) -> dx_colsum (fp16 or fp32)
__global__ __launch_bounds__(1024) void ln_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, void* __restrict__ dxsum, int dx_dtype,
                                                         int R, int N, int D) {
  // 32 columns x 32 row-lanes: only N / 32 blocks exist (72 for 3 x 768 columns), so each carries 16 waves to keep
  // enough 128-byte row segments in flight; the row-lanes are combined in fixed order (deterministic)
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  float a0 = 0.f, a1 = 0.f;
  if (col < N) {
    int r = rl;
    for (; r + 32 < R; r += 64) {
      a0 += partial[(long)r * N + col];
      a1 += partial[(long)(r + 32) * N + col];
    }
    if (r < R) a0 += partial[(long)r * N + col];
  }
  red[rl][c] = a0 + a1;
  __syncthreads();
  if (rl == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k][c];
    if (col < D) dgamma[col] = t;
    else if (col < 2 * D) dbeta[col - D] = t;
    else if (dx_dtype == 0) reinterpret_cast<half_t*>(dxsum)[col - 2 * D] = (half_t)t;
    else reinterpret_cast<float*>(dxsum)[col - 2 * D] = t;
  }
}

// ---- column sums: out[n] = sum_m X[m][n]  (two stages, fp32 partials) ---------------------------
// stage 1: block = 32 chunks (16 B each) x 8 row-lanes over a slab of rows; partial[slab][N]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, float* __restrict__ partial,
                                                             int M, int N, long ld, int rows_per_block) {
  constexpr int VN = Vec<T>::N;
  typedef typename Vec<T>::type V;
  __shared__ float red[8][32 * VN + 1];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cl) * VN;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float acc[VN];
#pragma unroll
  for (int j = 0; j < VN; ++j) acc[j] = 0.f;
  if (col < N) {
    int r = r0 + rl;
    for (; r + 8 < r1; r += 16) {
      V t0 = *reinterpret_cast<const V*>(X + (long)r * ld + col);
      V t1 = *reinterpret_cast<const V*>(X + (long)(r + 8) * ld + col);
#pragma unroll
      for (int j = 0; j < VN; ++j) acc[j] += (float)t0[j] + (float)t1[j];
    }
    if (r < r1) {
      V t0 = *reinterpret_cast<const V*>(X + (long)r * ld + col);
#pragma unroll
      for (int j = 0; j < VN; ++j) acc[j] += (float)t0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VN; ++j) red[rl][cl * VN + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * VN; i += 256) {
    int cc = blockIdx.x * 32 * VN + i;
    if (cc < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][i];
      partial[(long)blockIdx.y * N + cc] = t;
    }
  }
}

// 8 consecutive elements of an fp16 or fp32 row (the towers' as-written fp16 regime / the reference's fp32-upcast regime,
// modules/module_clip.py:566-577 followed by model.float()).  The fp32 forms keep every value unrounded.
__device__ __forceinline__ void store8(half_t* dst, const float (&v)[8]) {
  h8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
  *reinterpret_cast<h8*>(dst) = o;
}
__device__ __forceinline__ void store8(float* dst, const float (&v)[8]) {
  *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f4*>(dst + 4) = f4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void load8(const half_t* src, float (&v)[8]) {
  const h8 a = *reinterpret_cast<const h8*>(src);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
__device__ __forceinline__ void load8(const float* src, float (&v)[8]) {
  const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename T> struct is_f16 { static constexpr bool value = false; };
template <> struct is_f16<half_t> { static constexpr bool value = true; };
// the rounding a value takes when the reference holds it in a tensor of type T
template <typename T> __device__ __forceinline__ float rnd(float x) { return is_f16<T>::value ? r16(x) : x; }

// ---- patch extraction (im2col of the stride=patch conv), fp32 NCHW -> fp16 / fp32 [N*L, 3*p*p] ---------
// Row n*L + 0 is a zero row (class-token slot); row n*L + 1 + gy*g + gx is patch (gy, gx),
// columns ordered (c, ky, kx) like conv1.weight[width, 3, p, p].
template <typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, TO* __restrict__ out,
                                                       int nframes, int H, int W, int p, int g) {
  const int L = g * g + 1;
  const int kc = 3 * p * p / 8;                 // 16-byte chunks per output row
  const long total = (long)nframes * L * kc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int ch = (int)(idx % kc);
    long row = idx / kc;
    int l = (int)(row % L);
    long n = row / L;
    float o[8];
    if (l == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
    } else {
      int gy = (l - 1) / g, gx = (l - 1) % g;
      int per_c = p * p / 8;
      int c = ch / per_c, rem = ch % per_c;
      int ky = rem / (p / 8), kx = (rem % (p / 8)) * 8;
      load8(img + (((n * 3 + c) * H + gy * p + ky) * (long)W + gx * p + kx), o);
    }
    store8(out + idx * 8, o);
  }
}

// Same extraction from uint8 NCHW frames with the loader's normalisation fused in
// (dataloaders/dataloader_msrvtt_retrieval.py:242-247: ToTensor = x / 255 in fp32, Normalize = (x - mean[c]) / std[c] in
// fp32, then the model's image.type(fp16)): a quarter of the input bytes of the fp32 path.
struct PixNorm { float mean[3], std[3]; };
// frame_index (optional): output frame n is read from stored frame frame_index[n] (frame sampling, no gathered copy)
template <typename TO>
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ img, const int* __restrict__ frame_index,
                                                          TO* __restrict__ out, int nframes, int H, int W, int p, int g,
                                                          PixNorm nm) {
  const int L = g * g + 1;
  const int kc = 3 * p * p / 8;
  const long total = (long)nframes * L * kc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int ch = (int)(idx % kc);
    long row = idx / kc;
    int l = (int)(row % L);
    long n = row / L;
    float o[8];
    if (l == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
    } else {
      int gy = (l - 1) / g, gx = (l - 1) % g;
      int per_c = p * p / 8;
      int c = ch / per_c, rem = ch % per_c;
      int ky = rem / (p / 8), kx = (rem % (p / 8)) * 8;
      const long ns = frame_index ? (long)frame_index[n] : n;
      const unsigned char* src = img + (((ns * 3 + c) * H + gy * p + ky) * (long)W + gx * p + kx);
      const uint2 raw = *reinterpret_cast<const uint2*>(src);
      const float m = nm.mean[c], sd = nm.std[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned b = ((j < 4 ? raw.x : raw.y) >> (8 * (j & 3))) & 0xffu;
        o[j] = (((float)b / 255.0f) - m) / sd;
      }
    }
    store8(out + idx * 8, o);
  }
}

// x[n][l][:] = fp16( x[n][l][:] + (l == 0 ? fp16(cls) : 0) ) then fp16( . + fp16(pos[l]) )   (in place)
template <typename T>
__global__ __launch_bounds__(256) void vit_embed_kernel(T* __restrict__ x, const float* __restrict__ cls,
                                                        const float* __restrict__ pos, long rows, int L, int D) {
  const int dc = D / 8;
  const long total = rows * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % dc);
    long row = idx / dc;
    int l = (int)(row % L);
    float v[8];
    load8(x + idx * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j];
      if (l == 0) t = rnd<T>(cls[c * 8 + j]);           // conv output of the zero row is 0
      v[j] = rnd<T>(t + rnd<T>(pos[l * D + c * 8 + j]));
    }
    store8(x + idx * 8, v);
  }
}

// vit_embed_kernel and ln_pre in ONE pass over the patch-embed GEMM's output (fp16 tower; modules/module_clip.py:311-313): a wave
// per row reads x0, applies the class / positional embedding with the same rounding points, optionally writes the embedded
// row back (the backward of ln_pre needs it), normalises it in registers and writes y; `stat` (optional) receives
// (rstd, -rstd mean) of the ROUNDED y row for the first folded GEMM of the tower (ln_fold.hip).  Saves the 240 us in-place
// pass and the statistics pass at 153 600 x 768.  Structure of ln_fwd_kernel: next row requested before this one is reduced.
// FIXPOS: the grid's wave count is a multiple of L, so a wave meets ONE position all launch long: its positional row (3 KiB of
// fp32 per 1.5 KiB row of x0, from L2) is read once instead of per row (round 5: 162 -> see DESIGN.md section 6).
template <bool FIXPOS>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(half_t* __restrict__ x0, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, half_t* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           float* __restrict__ stat, int rows, int L, int D, float eps,
                                                           float stat_eps, int write_x0) {
  constexpr int MAXV = LN_MAXD / 64 / 8;
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int nchunk = D / 8;
  bool ok[MAXV];
  int ce[MAXV];
  float gm[MAXV][8], bt[MAXV][8], cl[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ok[i] = lane + 64 * i < nchunk;
    ce[i] = ok[i] ? (lane + 64 * i) * 8 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[i][j] = gamma[ce[i] + j]; bt[i][j] = beta[ce[i] + j]; cl[i][j] = r16(cls[ce[i] + j]); }
  }
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  h8 cur[MAXV], nxt[MAXV];
  f4 pc[MAXV][2], pn[MAXV][2];
  auto load_pos = [&](int r, f4 (&pp)[MAXV][2]) {
    const float* pr = pos + (long)(r % L) * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      pp[i][0] = *reinterpret_cast<const f4*>(pr + ce[i]);
      pp[i][1] = *reinterpret_cast<const f4*>(pr + ce[i] + 4);
    }
  };
  auto load_row = [&](int r, h8 (&t)[MAXV], f4 (&pp)[MAXV][2]) {
    const half_t* xr = x0 + (long)r * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) t[i] = *reinterpret_cast<const h8*>(xr + ce[i]);
    if constexpr (!FIXPOS) load_pos(r, pp);
  };
  load_row(row, cur, pc);
  if constexpr (FIXPOS) load_pos(row, pc);               // row % L is the same for every row this wave meets
  for (; row < rows; row += nwaves) {
    load_row(min(row + nwaves, rows - 1), nxt, pn);
    const bool is_cls = (row % L) == 0;
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      h8 e;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = is_cls ? cl[i][j] : (float)cur[i][j];           // conv output of the zero patch row is 0
        v[i][j] = r16(t + r16(pc[i][j >> 2][j & 3]));
        e[j] = (half_t)v[i][j];
        if (ok[i]) s += v[i][j];
      }
      if (write_x0 && ok[i]) *reinterpret_cast<h8*>(x0 + (long)row * D + ce[i]) = e;
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; if (ok[i]) q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / D + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    half_t* yr = y + (long)row * D;
    float ys = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = (half_t)((v[i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
        v[i][j] = (float)o[j];
        if (ok[i]) ys += v[i][j];
      }
      if (ok[i]) *reinterpret_cast<h8*>(yr + ce[i]) = o;
    }
    if (stat) {                                           // statistics of the row just written, two-pass like rowstat_kernel
      const float ym = wave_sum(ys) / D;
      float yq = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - ym; if (ok[i]) yq += d * d; }
      const float yr_ = 1.0f / sqrtf(wave_sum(yq) / D + stat_eps);
      if (lane == 0) *reinterpret_cast<f2*>(stat + 2 * (size_t)row) = f2{yr_, -yr_ * ym};
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      cur[i] = nxt[i];
      if constexpr (!FIXPOS) { pc[i][0] = pn[i][0]; pc[i][1] = pn[i][1]; }
    }
  }
}

// x[b][l][:] = fp16( fp16(table[ids[b][l]][:]) + fp16(pos[l][:]) )
// ids outside [0, vocab) never index the table: the row is written as the position embedding alone and *err is set
// (the reference's nn.Embedding raises an index error; a device kernel cannot, so the host checks the flag)
template <typename TO>
__global__ __launch_bounds__(256) void text_embed_kernel(const long* __restrict__ ids, const float* __restrict__ table,
                                                         const float* __restrict__ pos, TO* __restrict__ x,
                                                         long rows, int L, int D, long vocab, int* __restrict__ err) {
  const int dc = D / 8;
  const long total = rows * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % dc);
    long row = idx / dc;
    int l = (int)(row % L);
    const long id = ids[row];
    const bool ok = id >= 0 && id < vocab;
    if (!ok && c == 0 && err) *err = 1;
    const float* tr = table + (ok ? id : 0) * (long)D + c * 8;
    const float* pr = pos + (long)l * D + c * 8;
    f4 a = *reinterpret_cast<const f4*>(tr), b = *reinterpret_cast<const f4*>(tr + 4);
    if (!ok) { a = f4{0.f, 0.f, 0.f, 0.f}; b = a; }
    f4 pa = *reinterpret_cast<const f4*>(pr), pb = *reinterpret_cast<const f4*>(pr + 4);
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = rnd<TO>(a[j]) + rnd<TO>(pa[j]);
      o[4 + j] = rnd<TO>(b[j]) + rnd<TO>(pb[j]);
    }
    store8(x + idx * 8, o);
  }
}

// index[b] = base + b * stride + argmax_l ids[b][l] (first position of the largest id, as torch.argmax): the EOT row of caption b in
// a token-major [.., D] activation buffer (modules/module_cross.py:300-303 picks x[arange(b), text.argmax(-1)]).  One wave per
// caption; L <= 1024.
__global__ __launch_bounds__(256) void eot_index_kernel(const long* __restrict__ ids, int* __restrict__ index, int b, int L, long base,
                                                        long stride) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= b) return;
  long best = -0x7fffffffffffffffL - 1;
  int pos = 0;
  for (int l = lane; l < L; l += 64) {
    const long v = ids[(long)row * L + l];
    if (v > best) { best = v; pos = l; }           // strictly greater: a lane keeps its first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long ob = __shfl_xor(best, o, 64);
    const int op = __shfl_xor(pos, o, 64);
    if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
  }
  if (lane == 0) index[row] = (int)(base + (long)row * stride + pos);
}

// dtable[id][:] = sum over the rows r with ids[r] == id of dx[r][:]   (fp32 table gradient, dense, zeroed by the caller).
// Deterministic, no atomics: one workgroup per token row; the row that is the FIRST occurrence of its id owns the id's
// table row.  It lists the later occurrences in row order (wave ballots), its 8 waves each sum every 8th of them, and
// the 8 partial sums are added in wave order: a fixed summation tree whatever the launch timing.  Ids that occur once
// (most of them) cost one pass over the id list; the padding id's thousands of rows are 8 independent load chains.
constexpr int TE_THREADS = 512, TE_WAVES = 8, TE_LIST = 16384;
template <typename TI>
__global__ __launch_bounds__(TE_THREADS) void text_embed_bwd_kernel(const long* __restrict__ ids, const TI* __restrict__ dx,
                                                                    float* __restrict__ dtable, int rows, int D, long vocab) {
  __shared__ unsigned short list[TE_LIST];
  __shared__ int wave_cnt[TE_WAVES];
  __shared__ int base_cnt;
  __shared__ float part[TE_WAVES][512];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long my = ids[r];
  if (my < 0 || my >= vocab) return;                          // flagged by the forward pass; no gradient
  for (int base = 0; base < r; base += TE_THREADS) {          // an earlier row with this id owns it
    const int t = base + tid;
    if (__syncthreads_or(t < r && ids[t] == my)) return;
  }
  for (int c0 = 0; c0 < D; c0 += 512) {                       // column blocks of 512 (one per text width of CLIP)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int start = r;
    while (start < rows) {                                    // batches of at most TE_LIST occurrences
      if (tid == 0) base_cnt = 0;
      __syncthreads();
      int next = rows;
      for (int base = start; base < rows; base += TE_THREADS) {
        if (base + TE_THREADS - start > 65536) { next = base; break; }      // list entries are 16-bit offsets from `start` (block-uniform)
        const int t = base + tid;
        const bool hit = t < rows && ids[t] == my;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wid] = __popcll(m);
        __syncthreads();
        int off = base_cnt, tot = 0;
        for (int w = 0; w < TE_WAVES; ++w) { if (w < wid) off += wave_cnt[w]; tot += wave_cnt[w]; }
        const bool room = base_cnt + tot <= TE_LIST;          // block-uniform
        if (room && hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(t - start);
        __syncthreads();
        if (!room) { next = base; break; }
        if (tid == 0) base_cnt += tot;
        __syncthreads();
      }
      const int n = base_cnt;
      const int col = c0 + lane * 8;
      if (col < D) {
        for (int i = wid; i < n; i += TE_WAVES) {
          float v[8];
          load8(dx + (size_t)(start + list[i]) * D + col, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
      }
      __syncthreads();
      start = next;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[wid][lane * 8 + j] = acc[j];
    __syncthreads();
    if (c0 + tid < D) {
      float s = part[0][tid];
#pragma unroll
      for (int w = 1; w < TE_WAVES; ++w) s += part[w][tid];
      dtable[my * (long)D + c0 + tid] = s;
    }
    __syncthreads();
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (TO)in[i];
}

// ---- many second-stage reductions in one launch ---------------------------------------------------------------------------
// The backward of a tower leaves four fp32 partial matrices per layer (two LayerNorms: dgamma | dbeta | bias column sums;
// the c_fc bias partials of the QuickGELU' GEMM epilogue; the per-sequence in_proj bias sums of the attention backward).
// None of their sums is needed before the tower's backward returns, so instead of a 5 us reduce launch after each producer -
// ~100 launches per step on the critical path of the backward, 1.6 ms of a 17 ms step at 32 videos per GPU - the producers
// write into per-layer slots and ONE launch reduces all of them.  Task t: out_s[c % seg] = sum_r partial[r][c], s = c / seg.
struct ReduceTask { const float* partial; int R, N, seg; void* out[3]; int dtype[3]; };
constexpr int REDUCE_MAX_TASKS = 48;
struct ReduceArgs { ReduceTask t[REDUCE_MAX_TASKS]; int off[REDUCE_MAX_TASKS + 1]; int n; };

// A block takes 128 columns of one task (32 lanes x 4 columns: 512 contiguous bytes per row and 32-lane group) and its 8 row
// lanes walk the rows with four independent 16-byte loads in flight each (round 5; one column per thread and two loads in flight
// until then: 164 us for the 200 MB of partials of a b = 32 step).  Fixed order: row lane rl sums rows rl, rl + 8, ... in four
// interleaved accumulators (rows = rl + 8 (4 i + k) -> accumulator k), then k = 0..3, then the row lanes 0..7.
constexpr int MCR_COLS = 128;
__global__ __launch_bounds__(256) void multi_colreduce_kernel(ReduceArgs a) {
  __shared__ f4 red[8][33];
  int ti = 0;
  while (ti + 1 < a.n && (int)blockIdx.x >= a.off[ti + 1]) ++ti;           // block-uniform
  const ReduceTask& t = a.t[ti];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = ((int)blockIdx.x - a.off[ti]) * MCR_COLS + 4 * c;
  const float* P = t.partial;
  const int R = t.R, N = t.N;
  const bool vec = (N & 3) == 0 && col + 3 < N && ((uintptr_t)P & 15) == 0;    // 16-byte aligned rows: one f4 load per row
  f4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  auto load = [&](int r) -> f4 {
    const float* pr = P + (long)r * N + col;
    if (vec) return *reinterpret_cast<const f4*>(pr);
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) if (col + j < N) v[j] = pr[j];
    return v;
  };
  if (col < N) {
    int r = rl;
    for (; r + 24 < R; r += 32) {
      const f4 v0 = load(r), v1 = load(r + 8), v2 = load(r + 16), v3 = load(r + 24);
      acc[0] += v0; acc[1] += v1; acc[2] += v2; acc[3] += v3;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) if (r + 8 * k < R) acc[k] += load(r + 8 * k);
  }
  red[rl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (rl == 0 && col < N) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cj = col + j;
      if (cj >= N) break;
      const int sgm = cj / t.seg, pos = cj - sgm * t.seg;
      void* o = t.out[sgm];
      if (o) {
        if (t.dtype[sgm] == 0) reinterpret_cast<half_t*>(o)[pos] = (half_t)v[j];
        else reinterpret_cast<float*>(o)[pos] = v[j];
      }
    }
  }
}

inline int grid_for(long work_items, int block = 256, int cap = 2048) {
  long b = (work_items + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// dtype: 0 = fp16 I/O (CLIP towers), 1 = fp32 I/O (temporal transformer, MLM head)
extern "C" int hmmc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  const int* row_index, int rows, int D, long in_stride, float eps, int dtype,
                                  hipStream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return HMMC_ERR_ARG;
  int vn = dtype == 0 ? 8 : 4;
  if (D % vn || D > LN_MAXD || in_stride % vn) return HMMC_ERR_UNSUPPORTED;
  // 8 rows per wave at least (gamma / beta are loaded once per wave; 2 rows per wave up to 4 096 rows, where the chip is
  // otherwise idle and the rows of a wave are a latency chain), at most 8 workgroups per CU
  int nb = rows <= 4096 ? (rows + 7) / 8 : (rows + 31) / 32;
  const int cap = hmmc_num_cus() * 8;
  if (nb > cap) nb = cap;
  dim3 grid(nb < 1 ? 1 : nb), block(256);
  if (dtype == 0)
    hipLaunchKernelGGL(ln_fwd_kernel<half_t>, grid, block, 0, stream, (const half_t*)x, gamma, beta, (half_t*)y, mean, rstd,
                       row_index, rows, D, in_stride, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, stream, (const float*)x, gamma, beta, (float*)y, mean, rstd,
                       row_index, rows, D, in_stride, eps);
  return hmmc_launch_status();
}

// blocks of the LayerNorm backward = rows of its partial matrix: 8 rows per wave at least (the second-stage reduce reads
// nb x 3D floats, which at a few thousand rows would otherwise cost as much as the backward itself), and no more blocks
// than are resident at once (a second, partial round of blocks would leave most CUs idle at the end)
constexpr int LN_BWD_BLOCKS_PER_CU = 3;      // = the second __launch_bounds__ argument of ln_bwd_kernel: one resident round
static inline int ln_bwd_blocks(int rows) {
  // (up to 4 096 rows - the temporal transformer, the text tower at small batches - two rows per wave: at 384 rows 12 workgroups
  // walked 8 rows per wave one after the other on an otherwise idle chip, 20 us)
  int nb = rows <= 4096 ? (rows + 7) / 8 : (rows + 31) / 32;
  const int cap = hmmc_num_cus() * LN_BWD_BLOCKS_PER_CU;
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : nb;
}

extern "C" size_t hmmc_layernorm_bwd_workspace(int rows, int D) {
  return (size_t)ln_bwd_blocks(rows) * 3 * D * sizeof(float);
}

// rows of the partial matrix hmmc_layernorm_bwd_partial writes for `rows` input rows
extern "C" int hmmc_layernorm_bwd_rows(int rows) { return ln_bwd_blocks(rows); }

// First stage alone: dx as hmmc_layernorm_bwd, and `partial` = fp32 [hmmc_layernorm_bwd_rows(rows)][np * D] with np = 2
// (dgamma | dbeta partial sums per column) or 3 (... | column sums of dx) for want_dx_colsum; the caller reduces the rows later
// (hmmc_multi_colreduce, one launch for many such matrices).
extern "C" int hmmc_layernorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                          const void* dres, void* dx, float* partial, int want_dx_colsum, const int* row_index,
                                          int rows, int D, long in_stride, int dtype, hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !partial || rows <= 0) return HMMC_ERR_ARG;
  int vn = dtype == 0 ? 8 : 4;
  if (D % vn || D > LN_MAXD || in_stride % vn) return HMMC_ERR_UNSUPPORTED;
  const int nb = ln_bwd_blocks(rows);
  if (dtype == 0)
    hipLaunchKernelGGL(ln_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, stream, (const half_t*)dy, (const half_t*)x, gamma,
                       mean, rstd, (const half_t*)dres, (half_t*)dx, partial, row_index, rows, D, in_stride, want_dx_colsum != 0);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)dy, (const float*)x, gamma, mean,
                       rstd, (const float*)dres, (float*)dx, partial, row_index, rows, D, in_stride, want_dx_colsum != 0);
  return hmmc_launch_status();
}

// dx rows are written at the same (row_index, in_stride) positions the forward read x from.
// dx_colsum (optional, dtype of dx, [D]): sum over rows of the dx rows written, i.e. the bias gradient of the linear
// layer whose output this LayerNorm's input is; saves a separate pass over dx.
extern "C" int hmmc_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* dres, void* dx, float* dgamma, float* dbeta, void* dx_colsum,
                                  const int* row_index, int rows, int D, long in_stride, int dtype, void* workspace,
                                  size_t ws_bytes, hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0) return HMMC_ERR_ARG;
  int vn = dtype == 0 ? 8 : 4;
  if (D % vn || D > LN_MAXD || in_stride % vn) return HMMC_ERR_UNSUPPORTED;
  const int nb = ln_bwd_blocks(rows);
  const int np = dx_colsum ? 3 : 2;
  if (!workspace || ws_bytes < (size_t)nb * np * D * sizeof(float)) return HMMC_ERR_WORKSPACE;
  float* partial = (float*)workspace;
  if (dtype == 0)
    hipLaunchKernelGGL(ln_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, stream, (const half_t*)dy, (const half_t*)x, gamma,
                       mean, rstd, (const half_t*)dres, (half_t*)dx, partial, row_index, rows, D, in_stride, dx_colsum != nullptr);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)dy, (const float*)x, gamma, mean,
                       rstd, (const float*)dres, (float*)dx, partial, row_index, rows, D, in_stride, dx_colsum != nullptr);
  // one reduce over the np*D partial columns: [0,D) dgamma, [D,2D) dbeta (fp32), [2D,3D) dx_colsum (dtype of dx)
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((np * D + 31) / 32), dim3(1024), 0, stream, (const float*)partial, dgamma, dbeta,
                     dx_colsum, dtype, nb, np * D, D);
  return hmmc_launch_status();
}

static inline int colsum_slabs(int M) {
  int rb = (M + 63) / 64;              // >= 64 rows per slab
  if (rb > 1024) rb = 1024;
  return rb < 1 ? 1 : rb;
}

extern "C" size_t hmmc_colsum_workspace(int M, int N) { return (size_t)colsum_slabs(M) * N * sizeof(float); }

// out[n] = sum_m X[m][n]; in_dtype/out_dtype: 0 fp16, 1 fp32; round_f16: round the fp32 sum to fp16 first
extern "C" int hmmc_colsum(const void* X, void* out, int M, int N, long ld, int in_dtype, int out_dtype, int round_f16,
                           void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (!X || !out || M <= 0 || N <= 0) return HMMC_ERR_ARG;
  int vn = in_dtype == 0 ? 8 : 4;
  if (N % vn || ld % vn) return HMMC_ERR_UNSUPPORTED;
  // a few hundred dense fp32 rows (the partial sums a producer kernel left behind, at small batches): the second stage alone
  // reads them directly - one launch instead of two on the critical path of the backward pass
  if (in_dtype != 0 && ld == N && M <= 512) {
    if (out_dtype == 0)
      hipLaunchKernelGGL(colreduce_kernel<half_t>, dim3((N + 31) / 32), dim3(256), 0, stream, (const float*)X, (half_t*)out,
                         (half_t*)out, M, N, N, round_f16);
    else
      hipLaunchKernelGGL(colreduce_kernel<float>, dim3((N + 31) / 32), dim3(256), 0, stream, (const float*)X, (float*)out,
                         (float*)out, M, N, N, round_f16);
    return hmmc_launch_status();
  }
  int rb = colsum_slabs(M);
  int rpb = (M + rb - 1) / rb;
  rb = (M + rpb - 1) / rpb;
  if (!workspace || ws_bytes < (size_t)rb * N * sizeof(float)) return HMMC_ERR_WORKSPACE;
  dim3 grid((N / vn + 31) / 32, rb);
  if (in_dtype == 0)
    hipLaunchKernelGGL(colsum_partial_kernel<half_t>, grid, dim3(256), 0, stream, (const half_t*)X, (float*)workspace, M, N,
                       ld, rpb);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, (float*)workspace, M, N, ld,
                       rpb);
  if (out_dtype == 0)
    hipLaunchKernelGGL(colreduce_kernel<half_t>, dim3((N + 31) / 32), dim3(256), 0, stream, (const float*)workspace,
                       (half_t*)out, (half_t*)out, rb, N, N, round_f16);
  else
    hipLaunchKernelGGL(colreduce_kernel<float>, dim3((N + 31) / 32), dim3(256), 0, stream, (const float*)workspace,
                       (float*)out, (float*)out, rb, N, N, round_f16);
  return hmmc_launch_status();
}

extern "C" int hmmc_patchify(const float* img, void* out, int nframes, int H, int W, int patch, int out_dtype, hipStream_t stream) {
  if (!img || !out || nframes <= 0) return HMMC_ERR_ARG;
  if (patch % 8 || H % patch || W % patch || H != W || (W & 3)) return HMMC_ERR_UNSUPPORTED;
  int g = H / patch;
  long total = (long)nframes * (g * g + 1) * (3 * patch * patch / 8);
  if (out_dtype == 0)
    hipLaunchKernelGGL(patchify_kernel<half_t>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, img, (half_t*)out, nframes, H,
                       W, patch, g);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, img, (float*)out, nframes, H, W,
                       patch, g);
  return hmmc_launch_status();
}

extern "C" int hmmc_patchify_u8(const void* img, const int* frame_index, void* out, int nframes, int H, int W, int patch,
                                const float* mean3, const float* std3, int out_dtype, hipStream_t stream) {
  if (!img || !out || !mean3 || !std3 || nframes <= 0) return HMMC_ERR_ARG;      // mean3 / std3: HOST arrays of 3 floats
  if (patch % 8 || H % patch || W % patch || H != W || (W & 7) || ((uintptr_t)img & 7)) return HMMC_ERR_UNSUPPORTED;
  int g = H / patch;
  PixNorm nm;
  for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.std[c] = std3[c]; }
  long total = (long)nframes * (g * g + 1) * (3 * patch * patch / 8);
  if (out_dtype == 0)
    hipLaunchKernelGGL(patchify_u8_kernel<half_t>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, (const unsigned char*)img,
                       frame_index, (half_t*)out, nframes, H, W, patch, g, nm);
  else
    hipLaunchKernelGGL(patchify_u8_kernel<float>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, (const unsigned char*)img,
                       frame_index, (float*)out, nframes, H, W, patch, g, nm);
  return hmmc_launch_status();
}

extern "C" int hmmc_vit_embed(void* x, const float* cls, const float* pos, long rows, int L, int D, int dtype, hipStream_t stream) {
  if (!x || !cls || !pos || rows <= 0 || D % 8) return HMMC_ERR_ARG;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_embed_kernel<half_t>, dim3(grid_for(rows * (D / 8), 256, 4096)), dim3(256), 0, stream, (half_t*)x, cls,
                       pos, rows, L, D);
  else
    hipLaunchKernelGGL(vit_embed_kernel<float>, dim3(grid_for(rows * (D / 8), 256, 4096)), dim3(256), 0, stream, (float*)x, cls, pos,
                       rows, L, D);
  return hmmc_launch_status();
}

// fp16 tower only.  x0: the patch-embed GEMM output [rows, D] (class rows zero), rewritten with the embedded rows when write_x0;
// y = ln_pre(embedded) [rows, D]; mean / rstd [rows] of the embedded rows (for hmmc_layernorm_bwd); stat (optional, [rows][2]):
// (rstd, -rstd mean) of the rows of y, the operand hmmc_tower_fwd_fused's first folded GEMM needs.
extern "C" int hmmc_vit_embed_ln(void* x0, const float* cls, const float* pos, const float* gamma, const float* beta, void* y,
                                 float* mean, float* rstd, float* stat, int rows, int L, int D, float eps, int write_x0,
                                 hipStream_t stream) {
  if (!x0 || !cls || !pos || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || L <= 0) return HMMC_ERR_ARG;
  if (D % 8 || D > LN_MAXD) return HMMC_ERR_UNSUPPORTED;
  int nb = (rows + 31) / 32;
  const int cap = hmmc_num_cus() * 8;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  // a grid whose 4 nb waves are a multiple of L keeps every wave on one position: the largest such nb within the cap, if it
  // still fills the chip (ViT-B/32: L = 50 -> multiples of 25; ViT-B/16: L = 197 -> multiples of 197)
  const int unit = L / ((L % 4 == 0) ? 4 : (L % 2 == 0) ? 2 : 1);
  const int nbf = nb / unit * unit;
  if (nbf >= hmmc_num_cus() * 4) {
    hipLaunchKernelGGL(vit_embed_ln_kernel<true>, dim3(nbf), dim3(256), 0, stream, (half_t*)x0, cls, pos, gamma, beta, (half_t*)y,
                       mean, rstd, stat, rows, L, D, eps, eps, write_x0);
  } else {
    hipLaunchKernelGGL(vit_embed_ln_kernel<false>, dim3(nb), dim3(256), 0, stream, (half_t*)x0, cls, pos, gamma, beta, (half_t*)y,
                       mean, rstd, stat, rows, L, D, eps, eps, write_x0);
  }
  return hmmc_launch_status();
}

extern "C" int hmmc_text_embed(const long* ids, const float* table, const float* pos, void* x, long rows, int L, int D,
                               long vocab, int* err_flag, int out_dtype, hipStream_t stream) {
  if (!ids || !table || !pos || !x || rows <= 0 || D % 8 || vocab <= 0) return HMMC_ERR_ARG;
  if (out_dtype == 0)
    hipLaunchKernelGGL(text_embed_kernel<half_t>, dim3(grid_for(rows * (D / 8), 256, 4096)), dim3(256), 0, stream, ids, table, pos,
                       (half_t*)x, rows, L, D, vocab, err_flag);
  else
    hipLaunchKernelGGL(text_embed_kernel<float>, dim3(grid_for(rows * (D / 8), 256, 4096)), dim3(256), 0, stream, ids, table, pos,
                       (float*)x, rows, L, D, vocab, err_flag);
  return hmmc_launch_status();
}

extern "C" int hmmc_eot_index(const long* ids, int* index, int b, int L, long base, long stride, hipStream_t stream) {
  if (!ids || !index || b <= 0 || L <= 0 || L > 1024) return HMMC_ERR_ARG;
  if (base < 0 || stride < L || base + (long)b * stride > 0x7fffffffL) return HMMC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(eot_index_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, ids, index, b, L, base, stride);
  return hmmc_launch_status();
}

extern "C" int hmmc_text_embed_bwd(const long* ids, const void* dx, float* dtable, long rows, int D, long vocab, int dx_dtype,
                                   hipStream_t stream) {
  if (!ids || !dx || !dtable || rows <= 0 || vocab <= 0) return HMMC_ERR_ARG;
  if (D % 8 || rows > 0x7fffffffL) return HMMC_ERR_UNSUPPORTED;
  if (dx_dtype == 0)
    hipLaunchKernelGGL(text_embed_bwd_kernel<half_t>, dim3((unsigned)rows), dim3(TE_THREADS), 0, stream, ids, (const half_t*)dx,
                       dtable, (int)rows, D, vocab);
  else
    hipLaunchKernelGGL(text_embed_bwd_kernel<float>, dim3((unsigned)rows), dim3(TE_THREADS), 0, stream, ids, (const float*)dx, dtable,
                       (int)rows, D, vocab);
  return hmmc_launch_status();
}

// kind: 0 = fp16 -> fp32, 1 = fp32 -> fp16
extern "C" int hmmc_cast(const void* in, void* out, long n, int kind, hipStream_t stream) {
  if (!in || !out || n <= 0) return HMMC_ERR_ARG;
  if (kind == 0)
    hipLaunchKernelGGL((cast_kernel<half_t, float>), dim3(grid_for(n)), dim3(256), 0, stream, (const half_t*)in, (float*)out, n);
  else
    hipLaunchKernelGGL((cast_kernel<float, half_t>), dim3(grid_for(n)), dim3(256), 0, stream, (const float*)in, (half_t*)out, n);
  return hmmc_launch_status();
}

// tasks: HOST array of {partial, R, N, seg, out[3], dtype[3]} (layout of struct HmmcReduceTask in include/hmmc_hip.h)
extern "C" int hmmc_multi_colreduce(const void* tasks_host, int ntasks, hipStream_t stream) {
  if (!tasks_host || ntasks <= 0) return HMMC_ERR_ARG;
  const ReduceTask* tk = (const ReduceTask*)tasks_host;
  for (int base = 0; base < ntasks; base += REDUCE_MAX_TASKS) {
    ReduceArgs a;
    a.n = ntasks - base < REDUCE_MAX_TASKS ? ntasks - base : REDUCE_MAX_TASKS;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const ReduceTask& t = tk[base + i];
      if (!t.partial || t.R <= 0 || t.N <= 0 || t.seg <= 0 || (t.N + t.seg - 1) / t.seg > 3) return HMMC_ERR_ARG;
      a.t[i] = t;
      a.off[i] = blocks;
      blocks += (t.N + MCR_COLS - 1) / MCR_COLS;
    }
    a.off[a.n] = blocks;
    hipLaunchKernelGGL(multi_colreduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
  }
  return hmmc_launch_status();
}
