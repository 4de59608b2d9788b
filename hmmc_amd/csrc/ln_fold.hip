// LayerNorm folded into the GEMM that consumes it (fp16 CLIP towers, forward without saved activations).
//
// Reference: modules/module_clip.py:252-256 - x = x + attention(ln_1(x)); x = x + mlp(ln_2(x)) - with the fp32 LayerNorm of
// :217-223.  For a linear layer behind a LayerNorm,
//     y[r][n] = sum_k ((x[r][k] - mean_r) rstd_r gamma_k + beta_k) W[n][k] + b_n
//             = rstd_r (x W'^T)[r][n] - rstd_r mean_r c_n + d_n,    W' = gamma o W,  c_n = sum_k W'[n][k],  d_n = sum_k beta_k W[n][k] + b_n
// so the GEMM can read the RAW residual stream x and apply the row pair (rstd_r, -rstd_r mean_r) and the column pair
// (c_n, d_n) in its epilogue (HMMC_EPI_LNFOLD, gemm_f16.hip): LN(x) is never written to or read from HBM (2 x 472 MB per
// ViT-B/32 layer at 153 600 tokens).  Pieces here:
//   ln_fold_prep_kernel      W' (fp16, one rounding of gamma_k W[n][k]), c (sums of the ROUNDED W', so that the mean term cancels
//                            exactly what the MFMA accumulated) and d, for up to 32 weight matrices in one launch
//   rowstat_kernel           (rstd_r, -rstd_r mean_r) of fp16 rows: the tower's input (the residual stream of later layers gets
//                            its statistics from the producing GEMM's epilogue, HMMC_EPI_ROWSTAT)
//   rowstat_finalize_kernel  the per-64-column (sum, sum of squares) partials of HMMC_EPI_ROWSTAT -> (rstd_r, -rstd_r mean_r)
// Rounding differs from the unfolded path (gamma o W is rounded to fp16 instead of LN(x)): this is the opt-in / no-grad
// regime whose contract is the reference's own fp16-vs-fp32 envelope (tests/test_gpu_fold.py), not torch's rounding sequence.
#include "common.h"

namespace {

constexpr int PREP_MAX = 32;
struct PrepItem { const half_t* W; const float* gamma; const float* beta; const half_t* bias; half_t* Wf; float* cd; int N, row0; };
struct PrepArgs { PrepItem it[PREP_MAX]; int n, K, rows; };

// one wave per weight row
__global__ __launch_bounds__(256) void ln_fold_prep_kernel(PrepArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  int e = 0;
#pragma unroll 1
  for (int t = 1; t < a.n; ++t) e = row >= a.it[t].row0 ? t : e;
  const PrepItem& q = a.it[e];
  const int n = row - q.row0;
  const half_t* w = q.W + (size_t)n * a.K;
  half_t* wf = q.Wf + (size_t)n * a.K;
  float c = 0.f, d = 0.f;
  for (int k = lane * 8; k < a.K; k += 512) {
    const h8 wv = *reinterpret_cast<const h8*>(w + k);
    const f4 g0 = *reinterpret_cast<const f4*>(q.gamma + k), g1 = *reinterpret_cast<const f4*>(q.gamma + k + 4);
    const f4 b0 = *reinterpret_cast<const f4*>(q.beta + k), b1 = *reinterpret_cast<const f4*>(q.beta + k + 4);
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float wj = (float)wv[j];
      const float gj = j < 4 ? g0[j] : g1[j - 4], bj = j < 4 ? b0[j] : b1[j - 4];
      o[j] = (half_t)(gj * wj);
      c += (float)o[j];
      d += bj * wj;
    }
    *reinterpret_cast<h8*>(wf + k) = o;
  }
  c = wave_sum(c);
  d = wave_sum(d);
  if (lane == 0) {
    q.cd[n] = c;
    q.cd[q.N + n] = d + (q.bias ? (float)q.bias[n] : 0.f);
  }
}

// (rstd, -rstd * mean) of fp16 rows; two-pass statistics in registers as ln_fwd_kernel (norm_elem.hip), D <= 1024
__global__ __launch_bounds__(256) void rowstat_kernel(const half_t* __restrict__ x, float* __restrict__ stat, int rows, int D,
                                                       long stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  const int nchunk = D / 8;
  bool ok[2];
  int ce[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ok[i] = lane + 64 * i < nchunk; ce[i] = ok[i] ? (lane + 64 * i) * 8 : 0; }
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  h8 cur[2], nxt[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) cur[i] = *reinterpret_cast<const h8*>(x + (long)row * stride + ce[i]);
  for (; row < rows; row += nwaves) {
    const long nr = min(row + nwaves, rows - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) nxt[i] = *reinterpret_cast<const h8*>(x + nr * stride + ce[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) if (ok[i]) s += (float)cur[i][j];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dl = (float)cur[i][j] - mean; if (ok[i]) q += dl * dl; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / D + eps);
    if (lane == 0) *reinterpret_cast<f2*>(stat + 2 * (size_t)row) = f2{rstd, -rstd * mean};
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i] = nxt[i];
  }
}

// part: [nparts][rows][2] (sum, sum of squares) -> stat[rows][2] = (rstd, -rstd * mean)
__global__ __launch_bounds__(256) void rowstat_finalize_kernel(const float* __restrict__ part, float* __restrict__ stat, int nparts,
                                                                int rows, float inv_d, float eps) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f, q = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const f2 v = *reinterpret_cast<const f2*>(part + 2 * ((size_t)p * rows + r));
    s += v[0]; q += v[1];
  }
  const float mean = s * inv_d;
  const float var = fmaxf(q * inv_d - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + eps);
  *reinterpret_cast<f2*>(stat + 2 * (size_t)r) = f2{rstd, -rstd * mean};
}

}  // namespace

// W / gamma / beta / bias / Wf / cd / N: HOST arrays of `count` (<= 32) device pointers / row counts; every matrix is [N_e][K]
// fp16, K % 8 == 0, K <= 1024 x 8; bias entries may be NULL.  cd_e: [2][N_e] fp32 (c, then d).
extern "C" int hmmc_ln_fold_prep(const void* const* W, const float* const* gamma, const float* const* beta, const void* const* bias,
                                 void* const* Wf, float* const* cd, const int* N, int K, int count, hipStream_t stream) {
  if (!W || !gamma || !beta || !bias || !Wf || !cd || !N || count <= 0 || count > PREP_MAX || K <= 0) return HMMC_ERR_ARG;
  if (K & 7) return HMMC_ERR_UNSUPPORTED;
  PrepArgs a{};
  a.n = count; a.K = K;
  int rows = 0;
  for (int e = 0; e < count; ++e) {
    if (!W[e] || !gamma[e] || !beta[e] || !Wf[e] || !cd[e] || N[e] <= 0) return HMMC_ERR_ARG;
    if (((uintptr_t)W[e] | (uintptr_t)Wf[e] | (uintptr_t)gamma[e] | (uintptr_t)beta[e]) & 15) return HMMC_ERR_UNSUPPORTED;
    a.it[e] = PrepItem{(const half_t*)W[e], gamma[e], beta[e], (const half_t*)bias[e], (half_t*)Wf[e], cd[e], N[e], rows};
    rows += N[e];
  }
  a.rows = rows;
  hipLaunchKernelGGL(ln_fold_prep_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a);
  return hmmc_launch_status();
}

// stat[rows][2] = (rstd_r, -rstd_r mean_r) of the fp16 rows x[r * stride .. + D)
extern "C" int hmmc_rowstat(const void* x, float* stat, int rows, int D, long stride, float eps, hipStream_t stream) {
  if (!x || !stat || rows <= 0 || D <= 0) return HMMC_ERR_ARG;
  if ((D & 7) || D > 1024 || (stride & 7) || (((uintptr_t)x) & 15) || (((uintptr_t)stat) & 7)) return HMMC_ERR_UNSUPPORTED;
  const int cus = hmmc_num_cus();
  int blocks = (rows + 3) / 4;
  if (blocks > cus * 8) blocks = cus * 8;
  hipLaunchKernelGGL(rowstat_kernel, dim3(blocks), dim3(256), 0, stream, (const half_t*)x, stat, rows, D, stride, eps);
  return hmmc_launch_status();
}

// part: the [nparts][rows][2] output of a HMMC_EPI_ROWSTAT launch over rows of D = 64 nparts columns
extern "C" int hmmc_rowstat_finalize(const float* part, float* stat, int nparts, int rows, int D, float eps, hipStream_t stream) {
  if (!part || !stat || nparts <= 0 || rows <= 0 || D <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(rowstat_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, part, stat, nparts, rows, 1.0f / D, eps);
  return hmmc_launch_status();
}

// ---- backward of a folded LayerNorm --------------------------------------------------------------------------------------
// With y = u W'^T + d, u = (x - mean) rstd, W' = gamma o W (above), the data gradient that reaches the LayerNorm is
// du = dy W'.  The kernels in front of it hand over dy~ = rstd_r dy[r][:] (HMMC_EPI_ROWSCALE / the scaled attention backward),
// so the GEMM gives du~ = rstd_r du[r][:] and
//     dx[r][:] = du~ - mean_k(du~) - u[r][:] mean_k(du~ o u[r][:])  (+ the residual gradient)
// with no gamma in sight: ln_bwd_fold_kernel.  The weight gradient of the folded layer is taken against the RAW rows,
//     S[n][k] = sum_r dy~[r][n] x[r][k],    G[n][k] = sum_r dy~[r][n] (x[r][k] - mean_r) = S[n][k] - mean_k S[n][:]
// (sum_k (x[r][k] - mean_r) = 0, so the mean correction of every row of S is that row's own mean), and
//     dW = gamma_k G + beta_k db_n,   dgamma_k = sum_n W[n][k] G[n][k],   dbeta_k = sum_n W[n][k] db_n       (db = bias gradient)
// fold_grad_rowmean_kernel + fold_grad_finish_kernel, once per tower call for all layers.
namespace {

// structure of ln_bwd_kernel (norm_elem.hip) without gamma, dgamma, dbeta: a wave per row, next row requested early,
// per-block partial column sums of the dx rows written (the bias gradient of the layer that produced x)
__global__ __launch_bounds__(256, 4) void ln_bwd_fold_kernel(const half_t* __restrict__ dut, const half_t* __restrict__ x,
                                                              const float* __restrict__ stat, const half_t* __restrict__ dres,
                                                              half_t* __restrict__ dx, float* __restrict__ partial, int rows, int D,
                                                              long in_stride, int want_dxsum) {
  constexpr int MAXV = 2;                    // D <= 1024
  __shared__ float sred[4 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nchunk = D / 8;
  const int nwaves = gridDim.x * 4;
  bool ok[MAXV];
  int ce[MAXV];
  float ds[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ok[i] = lane + 64 * i < nchunk;
    ce[i] = ok[i] ? (lane + 64 * i) * 8 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) ds[i][j] = 0.f;
  }
  const bool has_res = dres != nullptr;
  auto load_row = [&](int row, h8 (&tx)[MAXV], h8 (&td)[MAXV], f2& st) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      tx[i] = *reinterpret_cast<const h8*>(x + (long)row * in_stride + ce[i]);
      td[i] = *reinterpret_cast<const h8*>(dut + (long)row * D + ce[i]);
    }
    st = *reinterpret_cast<const f2*>(stat + 2 * (size_t)row);
  };
  int row = blockIdx.x * 4 + w;
  h8 tx[MAXV], td[MAXV], tr[MAXV], nx[MAXV], nd[MAXV];
  f2 st = f2{0.f, 0.f}, nst;
  if (row < rows) load_row(row, tx, td, st);
  for (; row < rows; row += nwaves) {
    if (has_res) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) tr[i] = *reinterpret_cast<const h8*>(dres + (long)row * in_stride + ce[i]);
    }
    load_row(min(row + nwaves, rows - 1), nx, nd, nst);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)td[i][j];
          const float u = __builtin_fmaf(st[0], (float)tx[i][j], st[1]);
          s1 += d;
          s2 += d * u;
        }
      }
    }
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
    half_t* dxr = dx + (long)row * in_stride;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = __builtin_fmaf(st[0], (float)tx[i][j], st[1]);
        float val = (float)td[i][j] - s1 - u * s2;
        if (has_res) val += (float)tr[i][j];
        o[j] = (half_t)val;
        if (ok[i]) ds[i][j] += (float)o[j];
      }
      if (ok[i]) *reinterpret_cast<h8*>(dxr + ce[i]) = o;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { tx[i] = nx[i]; td[i] = nd[i]; }
    st = nst;
  }
  if (!want_dxsum) return;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (ok[i]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sred[w * 1024 + ce[i] + j] = ds[i][j];
    }
  }
  __syncthreads();
  for (int col = threadIdx.x; col < D; col += 256)
    partial[(long)blockIdx.x * D + col] = sred[col] + sred[1024 + col] + sred[2048 + col] + sred[3072 + col];
}

constexpr int FIN_MAX = 32;
constexpr int FIN_SLICES = 8;               // row slices per (matrix, 64 columns): 8 x the blocks, partial dgamma / dbeta per slice
struct FinItem { const float* S; const half_t* W; const float* gamma; const float* beta; const half_t* db; half_t* dW; float* dgamma; float* dbeta;
                 float* vmean; int N, row0, blk0; };
struct FinArgs { FinItem it[FIN_MAX]; int n, K, rows, blocks; };

// vmean[n] = mean_k S[n][k], one wave per row
__global__ __launch_bounds__(256) void fold_grad_rowmean_kernel(FinArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  int e = 0;
#pragma unroll 1
  for (int t = 1; t < a.n; ++t) e = row >= a.it[t].row0 ? t : e;
  const FinItem& q = a.it[e];
  const int n = row - q.row0;
  const float* sr = q.S + (size_t)n * a.K;
  float s = 0.f;
  for (int k = lane * 4; k < a.K; k += 256) {
    const f4 v = *reinterpret_cast<const f4*>(sr + k);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  s = wave_sum(s);
  if (lane == 0) q.vmean[n] = s / a.K;
}

// one block per (matrix, 64 columns, row slice): thread (column c, row group rg of 4) walks its slice's rows rg, rg + 4, ...
// eight at a time (eight independent row reads in flight per thread); the slice's dgamma / dbeta parts go to vmean[N + ...]
// (scratch [N + 2 FIN_SLICES K]) and fold_grad_sum_kernel adds the slices in order.
__global__ __launch_bounds__(256) void fold_grad_finish_kernel(FinArgs a) {
  __shared__ float red[2][4][64];
  int e = 0;
#pragma unroll 1
  for (int t = 1; t < a.n; ++t) e = (int)blockIdx.x >= a.it[t].blk0 ? t : e;
  const FinItem& q = a.it[e];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int local = (int)blockIdx.x - q.blk0;
  const int slice = local % FIN_SLICES;
  const int k = (local / FIN_SLICES) * 64 + c;
  const bool kok = k < a.K;
  const int kk = kok ? k : 0;
  const float gm = q.gamma[kk], bt = q.beta[kk];
  const int per = (q.N + FIN_SLICES - 1) / FIN_SLICES;
  const int n0 = slice * per, n1 = min(q.N, n0 + per);
  float dg = 0.f, dbt = 0.f;
  for (int nb = n0 + rg; nb < n1; nb += 32) {
    float Sv[8], Wv[8], vm[8], dbn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = min(nb + 4 * u, n1 - 1);
      Sv[u] = q.S[(size_t)n * a.K + kk];
      Wv[u] = (float)q.W[(size_t)n * a.K + kk];
      vm[u] = q.vmean[n];
      dbn[u] = (float)q.db[n];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = nb + 4 * u;
      if (n < n1) {
        const float G = Sv[u] - vm[u];
        if (kok) q.dW[(size_t)n * a.K + kk] = (half_t)(gm * G + bt * dbn[u]);
        dg += Wv[u] * G;
        dbt += Wv[u] * dbn[u];
      }
    }
  }
  red[0][rg][c] = dg;
  red[1][rg][c] = dbt;
  __syncthreads();
  if (rg == 0 && kok) {
    float* part = q.vmean + q.N + (size_t)slice * 2 * a.K;
    part[k] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    part[a.K + k] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

// dgamma / dbeta = the FIN_SLICES partial rows added in slice order (deterministic); one thread per (matrix, column)
__global__ __launch_bounds__(256) void fold_grad_sum_kernel(FinArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int e = idx / a.K, k = idx - e * a.K;
  if (e >= a.n) return;
  const FinItem& q = a.it[e];
  const float* part = q.vmean + q.N;
  float g = 0.f, b = 0.f;
#pragma unroll
  for (int sl = 0; sl < FIN_SLICES; ++sl) { g += part[(size_t)sl * 2 * a.K + k]; b += part[(size_t)sl * 2 * a.K + a.K + k]; }
  q.dgamma[k] = g;
  q.dbeta[k] = b;
}

}  // namespace

static inline int ln_bwd_fold_blocks(int rows) {
  int nb = (rows + 31) / 32;
  const int cap = hmmc_num_cus() * 4;                  // = the second __launch_bounds__ argument: one resident round
  return nb < 1 ? 1 : (nb > cap ? cap : nb);
}
// rows of the partial matrix [rows][D] hmmc_layernorm_bwd_fold writes when want_dx_colsum
extern "C" int hmmc_layernorm_bwd_fold_rows(int rows) { return ln_bwd_fold_blocks(rows); }

// dx[r] = du~[r] - mean(du~[r]) - u[r] mean(du~[r] o u[r]) (+ dres[r]), u = stat[r][0] x[r] + stat[r][1]; x / dres / dx rows at
// `stride`, du~ compact [rows][D]; partial (when want_dx_colsum): [hmmc_layernorm_bwd_fold_rows(rows)][D] fp32 column sums of dx
extern "C" int hmmc_layernorm_bwd_fold(const void* dut, const void* x, const float* stat, const void* dres, void* dx, float* partial,
                                       int want_dx_colsum, int rows, int D, long stride, hipStream_t stream) {
  if (!dut || !x || !stat || !dx || rows <= 0 || D <= 0 || (want_dx_colsum && !partial)) return HMMC_ERR_ARG;
  if ((D & 7) || D > 1024 || (stride & 7)) return HMMC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3(ln_bwd_fold_blocks(rows)), dim3(256), 0, stream, (const half_t*)dut, (const half_t*)x, stat,
                     (const half_t*)dres, (half_t*)dx, partial, rows, D, stride, want_dx_colsum);
  return hmmc_launch_status();
}

// The folded weight gradients of up to 32 matrices finished in two launches (HOST arrays of `count` entries): S_e fp32 [N_e][K]
// (sums against the raw rows), W_e fp16 [N_e][K], gamma_e / beta_e fp32 [K], db_e fp16 [N_e] (the layer's bias gradient) ->
// dW_e fp16 [N_e][K], dgamma_e / dbeta_e fp32 [K]; vmean_e: fp32 [hmmc_fold_grad_scratch_floats(N_e, K)] scratch.
extern "C" size_t hmmc_fold_grad_scratch_floats(int N, int K) { return (size_t)N + 2 * (size_t)FIN_SLICES * K; }

extern "C" int hmmc_fold_grad_finish(const float* const* S, const void* const* W, const float* const* gamma, const float* const* beta,
                                     const void* const* db, void* const* dW, float* const* dgamma, float* const* dbeta,
                                     float* const* vmean, const int* N, int K, int count, hipStream_t stream) {
  if (!S || !W || !gamma || !beta || !db || !dW || !dgamma || !dbeta || !vmean || !N || count <= 0 || count > FIN_MAX || K <= 0) return HMMC_ERR_ARG;
  if (K & 3) return HMMC_ERR_UNSUPPORTED;
  FinArgs a{};
  a.n = count; a.K = K;
  int rows = 0, blocks = 0;
  const int kb = (K + 63) / 64;
  for (int e = 0; e < count; ++e) {
    if (!S[e] || !W[e] || !gamma[e] || !beta[e] || !db[e] || !dW[e] || !dgamma[e] || !dbeta[e] || !vmean[e] || N[e] <= 0) return HMMC_ERR_ARG;
    a.it[e] = FinItem{S[e], (const half_t*)W[e], gamma[e], beta[e], (const half_t*)db[e], (half_t*)dW[e], dgamma[e], dbeta[e], vmean[e], N[e], rows, blocks};
    rows += N[e];
    blocks += kb * FIN_SLICES;
  }
  a.rows = rows; a.blocks = blocks;
  hipLaunchKernelGGL(fold_grad_rowmean_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(fold_grad_finish_kernel, dim3(blocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(fold_grad_sum_kernel, dim3((count * K + 255) / 256), dim3(256), 0, stream, a);
  return hmmc_launch_status();
}
