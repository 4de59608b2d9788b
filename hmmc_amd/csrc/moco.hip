// fp32 kernels of the pre-training (MoCo) heads, all HBM/launch-bound:
//   * BatchNorm1d of the projector / predictor MLPs (train-mode batch statistics, SyncBN-ready:
//     sums are produced per rank and all-reduced by the caller)        reference modules/modeling.py:788-807
//   * contrastive_loss against a negative queue, batched over every call that shares a queue
//     (the reference runs 48 separate calls and clones the queue each time)          modules/modeling.py:286-332
//   * MLM head pieces: erf-GELU and cross-entropy with ignore_index                  modules/module_cross.py:33-39,
//                                                                                    modules/modeling.py:171-179
// The logits S = qn . queue come from hmmc_gemm_f32 (queue read once per group, never copied).
#include "common.h"

namespace {

inline int nblk(long work, int cap = 4096) {
  long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---------------------------------------------------------------- BatchNorm1d over rows
// partial[slab][0|1][N]: per-slab column sum and sum of squares; 32 columns x 8 row lanes per block
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ h, float* __restrict__ partial,
                                                               int M, int N, int rows_per_block) {
  __shared__ float red[2][8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f, q = 0.f;
  if (col < N)
    for (int r = r0 + rl; r < r1; r += 8) { float v = h[(long)r * N + col]; s += v; q += v * v; }
  red[0][rl][c] = s; red[1][rl][c] = q;
  __syncthreads();
  if (rl < 2 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[rl][k][c];
    partial[((long)blockIdx.y * 2 + rl) * N + col] = t;
  }
}

// sums[0][n] = sum_slab partial[slab][0][n], sums[1][n] likewise
__global__ __launch_bounds__(256) void bn_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ sums,
                                                                int slabs, int N) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * N) return;
  int which = idx / N, col = idx - which * N;
  float t = 0.f;
  for (int s = 0; s < slabs; ++s) t += partial[((long)s * 2 + which) * N + col];
  sums[idx] = t;
}

// y = relu((h - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ h, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, long M,
                                                            int N) {
  const long total = M * N / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int col = (int)((i * 4) % N);
    f4 v = *reinterpret_cast<const f4*>(h + i * 4), o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf((v[j] - mean[col + j]) * rstd[col + j] * gamma[col + j] + beta[col + j], 0.f);
    *reinterpret_cast<f4*>(y + i * 4) = o;
  }
}

// d = dy * (y > 0); partial[slab][0][n] = sum d, partial[slab][1][n] = sum d * xhat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ h, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ partial,
                                                             int M, int N, int rows_per_block) {
  __shared__ float red[2][8][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f, q = 0.f;
  if (col < N) {
    const float mu = mean[col], rs = rstd[col];
    for (int r = r0 + rl; r < r1; r += 8) {
      long o = (long)r * N + col;
      float d = y[o] > 0.f ? dy[o] : 0.f;
      s += d; q += d * (h[o] - mu) * rs;
    }
  }
  red[0][rl][c] = s; red[1][rl][c] = q;
  __syncthreads();
  if (rl < 2 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[rl][k][c];
    partial[((long)blockIdx.y * 2 + rl) * N + col] = t;
  }
}

// Batch statistics of BatchNorm1d from the column sums, and the running statistics of train mode, in one launch (round 5: the
// host side spent 14 elementwise launches on these [N]-sized vectors per MLP call): mean = s1 / n, var = max(s2 / n - mean^2, 0)
// (biased: what normalises), rstd = rsqrt(var + eps); running_mean = (1 - mom) running_mean + mom mean, running_var likewise
// with the unbiased var n / (n - 1) (torch.nn.BatchNorm1d), num_batches_tracked += 1.  n = *n_dev (the all-reduced row count of
// SyncBatchNorm) or n_host.  The arithmetic follows the tensor expressions op by op (separate roundings, no fused contraction).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ n_dev, float n_host,
                                                          float eps, float mom, float* __restrict__ mean, float* __restrict__ var,
                                                          float* __restrict__ rstd, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, long* __restrict__ nbt, int N) {
  const float n = n_dev ? n_dev[0] : n_host;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col == 0 && nbt) nbt[0] += 1;
  if (col >= N) return;
  const float m = sums[col] / n;
  const float v = fmaxf(opq(sums[N + col] / n) - opq(m * m), 0.0f);
  mean[col] = m;
  var[col] = v;
  rstd[col] = 1.0f / sqrtf(v + eps);              // torch.rsqrt on the GPU: correctly rounded 1 / sqrt here, within 1 ulp of it
  if (run_mean) {
    const float nm1 = fmaxf(n - 1.0f, 1.0f);
    run_mean[col] = opq(run_mean[col] * (1.0f - mom)) + opq(mom * m);
    run_var[col] = opq(run_var[col] * (1.0f - mom)) + opq(mom * opq(v * opq(n / nm1)));
  }
}

// dh = gamma * rstd * (d - sum_d/n - xhat * sum_dx/n), d = dy * (y > 0); n = global row count
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ h, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float* __restrict__ dh, long M,
                                                           int N, float inv_n) {
  const long total = M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int col = (int)(i % N);
    float d = y[i] > 0.f ? dy[i] : 0.f;
    float xh = (h[i] - mean[col]) * rstd[col];
    dh[i] = gamma[col] * rstd[col] * (d - sums[col] * inv_n - xh * sums[N + col] * inv_n);
  }
}

// ---------------------------------------------------------------- contrastive loss against a queue
// out[r] = <a[r], b[r]>
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += a[(long)row * D + c] * b[(long)row * D + c];
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// per row: lse = log( exp(lpos/T) + sum_n exp(S[r][n]/T) ); rowloss = w * (lse - lpos/T).  One block per row.
__global__ __launch_bounds__(256) void moco_lse_kernel(const float* __restrict__ S, const float* __restrict__ lpos,
                                                       float* __restrict__ lse, float* __restrict__ rowloss, long Kq,
                                                       float invT, float w) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* sr = S + (long)r * Kq;
  const float lp = lpos[r] * invT;
  float m = lp;
  for (long n = threadIdx.x; n < Kq; n += 256) m = fmaxf(m, sr[n] * invT);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (long n = threadIdx.x; n < Kq; n += 256) s += __expf(sr[n] * invT - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = red[0] + red[1] + red[2] + red[3] + __expf(lp - m);
    float l = m + __logf(tot);
    lse[r] = l;
    rowloss[r] = w * (l - lp);
  }
}

// loss = sum_r rowloss[r]  (single block, deterministic)
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float red[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// dS[r][n] = g*w/T * exp(S/T - lse) (in place over S); dlpos[r] = g*w/T * (exp(lpos/T - lse) - 1)
__global__ __launch_bounds__(256) void moco_bwd_kernel(float* __restrict__ S, const float* __restrict__ lpos,
                                                       const float* __restrict__ lse, const float* __restrict__ gout,
                                                       float* __restrict__ dlpos, int R, long Kq, float invT, float w) {
  const float g = gout[0] * w * invT;
  const long total = (long)R * Kq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / Kq);
    S[i] = g * __expf(S[i] * invT - lse[r]);
    if (i - (long)r * Kq == 0) dlpos[r] = g * (__expf(lpos[r] * invT - lse[r]) - 1.0f);
  }
}

// y[r][:] += s[r] * x[r][:]
__global__ __launch_bounds__(256) void row_axpy_kernel(float* __restrict__ y, const float* __restrict__ s,
                                                       const float* __restrict__ x, long rows, int D) {
  const long total = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    y[i] += s[i / D] * x[i];
}

// ---------------------------------------------------------------- MLM head pieces
__global__ __launch_bounds__(256) void gelu_erf_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = x[i] * 0.5f * (1.0f + erff(x[i] * 0.70710678118654752f));
}
__global__ __launch_bounds__(256) void gelu_erf_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}

// cross entropy with ignore_index < 0: per row lse and loss contribution; one block per row
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long* __restrict__ labels,
                                                     float* __restrict__ lse, float* __restrict__ rowloss, long V) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const long lab = labels[r];
  if (lab < 0) { if (threadIdx.x == 0) { lse[r] = 0.f; rowloss[r] = 0.f; } return; }
  const float* lr = logits + (long)r * V;
  float m = -INFINITY;
  for (long n = threadIdx.x; n < V; n += 256) m = fmaxf(m, lr[n]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (long n = threadIdx.x; n < V; n += 256) s += __expf(lr[n] - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = m + __logf(red[0] + red[1] + red[2] + red[3]);
    lse[r] = l;
    rowloss[r] = l - lr[lab];
  }
}

// count[0] = number of rows with label >= 0
__global__ __launch_bounds__(256) void count_valid_kernel(const long* __restrict__ labels, float* __restrict__ count, int R) {
  __shared__ float red[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) a += labels[i] >= 0 ? 1.f : 0.f;
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = red[0];
}

// dlogits (in place) = (softmax - onehot) * g / n_valid for valid rows, 0 otherwise
__global__ __launch_bounds__(256) void ce_bwd_kernel(float* __restrict__ logits, const long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ gout,
                                                     const float* __restrict__ count, int R, long V) {
  const float g = gout[0] / fmaxf(count[0], 1.0f);
  const long total = (long)R * V;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / V);
    long lab = labels[r];
    float o = 0.f;
    if (lab >= 0) {
      o = __expf(logits[i] - lse[r]);
      if (i - (long)r * V == lab) o -= 1.0f;
      o *= g;
    }
    logits[i] = o;
  }
}

}  // namespace

static inline int bn_slabs(int M) { int s = (M + 63) / 64; return s > 256 ? 256 : (s < 1 ? 1 : s); }

extern "C" size_t hmmc_bn_workspace(int M, int N) { return (size_t)bn_slabs(M) * 2 * N * sizeof(float); }

// sums[0][n] = sum_m h[m][n]; sums[1][n] = sum_m h[m][n]^2   (this rank's rows; the caller all-reduces for SyncBN)
extern "C" int hmmc_bn_stats(const float* h, float* sums, int M, int N, void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (!h || !sums || M <= 0 || N <= 0) return HMMC_ERR_ARG;
  int slabs = bn_slabs(M), rpb = (M + slabs - 1) / slabs;
  slabs = (M + rpb - 1) / rpb;
  if (!workspace || ws_bytes < (size_t)slabs * 2 * N * sizeof(float)) return HMMC_ERR_WORKSPACE;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((N + 31) / 32, slabs), dim3(256), 0, stream, h, (float*)workspace, M, N, rpb);
  hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((2 * N + 255) / 256), dim3(256), 0, stream, (const float*)workspace, sums,
                     slabs, N);
  return hmmc_launch_status();
}

extern "C" int hmmc_bn_apply_relu(const float* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  float* y, long M, int N, hipStream_t stream) {
  if (!h || !mean || !rstd || !gamma || !beta || !y || M <= 0 || (N & 3)) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(nblk(M * N / 4)), dim3(256), 0, stream, h, mean, rstd, gamma, beta, y, M, N);
  return hmmc_launch_status();
}

// sums[0][n] = sum d, sums[1][n] = sum d * xhat, d = dy * (y > 0)
extern "C" int hmmc_bn_bwd_reduce(const float* dy, const float* y, const float* h, const float* mean, const float* rstd,
                                  float* sums, int M, int N, void* workspace, size_t ws_bytes, hipStream_t stream) {
  if (!dy || !y || !h || !mean || !rstd || !sums || M <= 0 || N <= 0) return HMMC_ERR_ARG;
  int slabs = bn_slabs(M), rpb = (M + slabs - 1) / slabs;
  slabs = (M + rpb - 1) / rpb;
  if (!workspace || ws_bytes < (size_t)slabs * 2 * N * sizeof(float)) return HMMC_ERR_WORKSPACE;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((N + 31) / 32, slabs), dim3(256), 0, stream, dy, y, h, mean, rstd,
                     (float*)workspace, M, N, rpb);
  hipLaunchKernelGGL(bn_partial_reduce_kernel, dim3((2 * N + 255) / 256), dim3(256), 0, stream, (const float*)workspace, sums,
                     slabs, N);
  return hmmc_launch_status();
}

extern "C" int hmmc_bn_bwd_apply(const float* dy, const float* y, const float* h, const float* mean, const float* rstd,
                                 const float* gamma, const float* sums, float* dh, long M, int N, float inv_n,
                                 hipStream_t stream) {
  if (!dy || !y || !h || !mean || !rstd || !gamma || !sums || !dh || M <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(M * N)), dim3(256), 0, stream, dy, y, h, mean, rstd, gamma, sums, dh, M, N,
                     inv_n);
  return hmmc_launch_status();
}

extern "C" int hmmc_bn_finalize(const float* sums, const float* n_dev, float n_host, float eps, float momentum, float* mean,
                                float* var, float* rstd, float* running_mean, float* running_var, long* num_batches_tracked, int N,
                                hipStream_t stream) {
  if (!sums || !mean || !var || !rstd || N <= 0 || (!n_dev && !(n_host > 0.f)) || (!running_mean != !running_var)) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, sums, n_dev, n_host, eps, momentum, mean, var, rstd,
                     running_mean, running_var, num_batches_tracked, N);
  return hmmc_launch_status();
}

extern "C" int hmmc_rowdot(const float* a, const float* b, float* out, int rows, int D, hipStream_t stream) {
  if (!a || !b || !out || rows <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(rowdot_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a, b, out, rows, D);
  return hmmc_launch_status();
}

// loss = sum_r w * (logsumexp([lpos[r], S[r][:]] / T) - lpos[r]/T); lse[R] saved; rowloss[R] scratch
extern "C" int hmmc_moco_loss_fwd(const float* S, const float* lpos, float* lse, float* rowloss, float* loss, int R, long Kq,
                                  float temperature, float w, hipStream_t stream) {
  if (!S || !lpos || !lse || !rowloss || !loss || R <= 0 || Kq <= 0 || temperature <= 0.f) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(moco_lse_kernel, dim3(R), dim3(256), 0, stream, S, lpos, lse, rowloss, Kq, 1.0f / temperature, w);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)rowloss, loss, R);
  return hmmc_launch_status();
}

// S is overwritten with dS; dlpos[R]
extern "C" int hmmc_moco_loss_bwd(float* S, const float* lpos, const float* lse, const float* grad_out, float* dlpos, int R,
                                  long Kq, float temperature, float w, hipStream_t stream) {
  if (!S || !lpos || !lse || !grad_out || !dlpos || R <= 0 || Kq <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(moco_bwd_kernel, dim3(nblk((long)R * Kq, 8192)), dim3(256), 0, stream, S, lpos, lse, grad_out, dlpos, R, Kq,
                     1.0f / temperature, w);
  return hmmc_launch_status();
}

extern "C" int hmmc_row_axpy(float* y, const float* s, const float* x, long rows, int D, hipStream_t stream) {
  if (!y || !s || !x || rows <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(row_axpy_kernel, dim3(nblk(rows * D)), dim3(256), 0, stream, y, s, x, rows, D);
  return hmmc_launch_status();
}

extern "C" int hmmc_gelu_erf_fwd(const float* x, float* y, long n, hipStream_t stream) {
  if (!x || !y || n <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(gelu_erf_fwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, y, n);
  return hmmc_launch_status();
}

extern "C" int hmmc_gelu_erf_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t stream) {
  if (!x || !dy || !dx || n <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(gelu_erf_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, dy, dx, n);
  return hmmc_launch_status();
}

// mean over rows with label >= 0 of (lse - logit[label]) (F.cross_entropy, ignore_index=-100); count[0] = #valid rows
extern "C" int hmmc_ce_fwd(const float* logits, const long* labels, float* lse, float* rowloss, float* count, float* loss,
                           int R, long V, hipStream_t stream) {
  if (!logits || !labels || !lse || !rowloss || !count || !loss || R <= 0 || V <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(R), dim3(256), 0, stream, logits, labels, lse, rowloss, V);
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, stream, labels, count, R);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, stream, (const float*)rowloss, loss, R);
  return hmmc_launch_status();
}

// logits is overwritten with d loss_sum / d logits * grad_out / count
extern "C" int hmmc_ce_bwd(float* logits, const long* labels, const float* lse, const float* grad_out, const float* count,
                           int R, long V, hipStream_t stream) {
  if (!logits || !labels || !lse || !grad_out || !count || R <= 0 || V <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(nblk((long)R * V, 8192)), dim3(256), 0, stream, logits, labels, lse, grad_out, count, R, V);
  return hmmc_launch_status();
}
