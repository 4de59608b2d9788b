// fp32 heads of the HMMC hot path (HBM/launch-bound, K8-K10 of SURVEY.md section 2.3):
//   * row L2-normalisation fwd/bwd                      (modules/modeling.py:210-214, F.normalize at :289-292)
//   * hierarchical InfoNCE over the (B x B) video-text and F (B x B) frame-text matrices,
//     CrossEn on S and S^T                               (modules/modeling.py:665-672,702-709; until_module.py:196-205)
//   * eval-time top-k-frames mean                        (main_task_retrieval.py:332-336)
//   * temporal pooling  mean_f( (h+u)/||h+u|| )          (modules/module_cross.py:207-212)
//   * the temporal transformer's F x F fp32 attention    (modules/module_cross.py:127-131)
// The similarity logits themselves come from hmmc_gemm_f32 as ONE [B, B*(1+F)] matrix
// S_all = 100 * Qn [Vn ; Un]^T: column c < B is video c, column B + b*F + f is frame f of video b,
// so the reference's F+1 separate matmuls and 2(F+1) log-softmax passes become one GEMM and three
// small kernels; nothing is re-normalised F times.
#include "common.h"

namespace {

// ---------------------------------------------------------------- L2 normalise rows (D % 4 == 0)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    f4 v = *reinterpret_cast<const f4*>(xr + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  float n = sqrtf(wave_sum(s));
  float d = fmaxf(n, eps);
  if (lane == 0) norm[row] = d;
  float inv = 1.0f / d;
  for (int c = lane * 4; c < D; c += 256) {
    f4 v = *reinterpret_cast<const f4*>(xr + c);
    *reinterpret_cast<f4*>(y + (long)row * D + c) = v * inv;
  }
}

// dx = (dy - y * <y, dy>) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norm, float* __restrict__ dx,
                                                         int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* yr = y + (long)row * D;
  const float* dr = dy + (long)row * D;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    f4 a = *reinterpret_cast<const f4*>(yr + c), b = *reinterpret_cast<const f4*>(dr + c);
    s += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  s = wave_sum(s);
  float inv = 1.0f / norm[row];
  for (int c = lane * 4; c < D; c += 256) {
    f4 a = *reinterpret_cast<const f4*>(yr + c), b = *reinterpret_cast<const f4*>(dr + c);
    *reinterpret_cast<f4*>(dx + (long)row * D + c) = (b - a * s) * inv;
  }
}

// ---------------------------------------------------------------- hierarchical InfoNCE
__device__ __forceinline__ int col_of(int b, int blk, int B, int F) { return blk == 0 ? b : B + b * F + (blk - 1); }

// row LSE: lse_row[i][blk] = log sum_b exp(S[i][col(b, blk)])
__global__ __launch_bounds__(64) void infonce_rowlse_kernel(const float* __restrict__ S, float* __restrict__ lse_row,
                                                            int B, int F) {
  const int i = blockIdx.x, blk = blockIdx.y, lane = threadIdx.x;
  const long C = (long)B * (1 + F);
  const float* sr = S + (long)i * C;
  float m = -INFINITY;
  for (int b = lane; b < B; b += 64) m = fmaxf(m, sr[col_of(b, blk, B, F)]);
  m = wave_max(m);
  float s = 0.f;
  for (int b = lane; b < B; b += 64) s += __expf(sr[col_of(b, blk, B, F)] - m);
  s = wave_sum(s);
  if (lane == 0) lse_row[(long)i * (1 + F) + blk] = m + __logf(s);
}

// column LSE: lse_col[c] = log sum_i exp(S[i][c]).  A block takes 32 columns; its 8 row lanes run the online max / sum over rows
// rl, rl + 8, ... (128-byte coalesced per 32-lane group) and their (max, sum) pairs are merged in row-lane order - a fixed order,
// so the result is bit-stable.  (Round 5: one thread per column walked all B rows in one dependent chain - 72 us for the
// 256 x 3 328 matrix of config 2 on 13 workgroups.)
__global__ __launch_bounds__(256) void infonce_collse_kernel(const float* __restrict__ S, float* __restrict__ lse_col,
                                                             int B, long C) {
  __shared__ float sm[8][32], ss[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long c = (long)blockIdx.x * 32 + cl;
  float m = -INFINITY, s = 0.f;
  if (c < C) {
    for (int i = rl; i < B; i += 8) {
      const float v = S[(long)i * C + c];
      if (v > m) { s = s * __expf(m - v) + 1.0f; m = v; }
      else s += __expf(v - m);
    }
  }
  sm[rl][cl] = m; ss[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float M = sm[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) M = fmaxf(M, sm[k][cl]);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += ss[k][cl] > 0.f ? ss[k][cl] * __expf(sm[k][cl] - M) : 0.f;     // a row lane without rows: (-inf, 0)
    lse_col[c] = M + __logf(t);
  }
}

// loss = sum_blk w_blk/B * [ sum_i (lse_row[i][blk] - S[i][col(i,blk)]) + sum_b (lse_col[col(b,blk)] - S[b][col(b,blk)]) ]
__global__ __launch_bounds__(256) void infonce_loss_kernel(const float* __restrict__ S, const float* __restrict__ lse_row,
                                                           const float* __restrict__ lse_col, float* __restrict__ loss,
                                                           int B, int F, float w_video, float w_frame) {
  __shared__ float red[256];
  const long C = (long)B * (1 + F);
  float acc = 0.f;
  for (int idx = threadIdx.x; idx < B * (1 + F); idx += 256) {
    int i = idx / (1 + F), blk = idx % (1 + F);
    float w = blk == 0 ? w_video : w_frame;
    int c = col_of(i, blk, B, F);
    float d = S[(long)i * C + c];
    acc += w * ((lse_row[idx] - d) + (lse_col[c] - d));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / B;
}

// dS[i][c] = g * w_blk/B * ( softmax_row + softmax_col - 2*[i == b(c)] )
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ S, const float* __restrict__ lse_row,
                                                          const float* __restrict__ lse_col, const float* __restrict__ gout,
                                                          float* __restrict__ dS, int B, int F, float w_video,
                                                          float w_frame) {
  const long C = (long)B * (1 + F);
  const long total = (long)B * C;
  const float g = gout[0] / B;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int i = (int)(idx / C);
    long c = idx - (long)i * C;
    int b, blk;
    if (c < B) { b = (int)c; blk = 0; }
    else { b = (int)((c - B) / F); blk = 1 + (int)((c - B) % F); }
    float w = blk == 0 ? w_video : w_frame;
    float v = S[idx];
    float t = __expf(v - lse_row[(long)i * (1 + F) + blk]) + __expf(v - lse_col[c]) - (i == b ? 2.0f : 0.0f);
    dS[idx] = g * w * t;
  }
}

// out[i][b] = (base ? base[i*ldb + b] : 0) + mean of the k largest of Sf[i*lds + b*F + f], f < F (F <= 64)
__global__ __launch_bounds__(256) void topk_mean_kernel(const float* __restrict__ Sf, const float* __restrict__ base,
                                                        float* __restrict__ out, int bq, int bv, int F, int k, long lds,
                                                        long ldb) {
  const long total = (long)bq * bv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int i = (int)(idx / bv), b = (int)(idx % bv);
    const float* v = Sf + (long)i * lds + (long)b * F;
    unsigned long long used = 0ull;
    float sum = 0.f;
    for (int t = 0; t < k; ++t) {
      float best = -INFINITY;
      int bi = 0;
      for (int f = 0; f < F; ++f) {
        float x = v[f];
        if (!((used >> f) & 1ull) && x > best) { best = x; bi = f; }
      }
      used |= 1ull << bi;
      sum += best;
    }
    out[idx] = (base ? base[(long)i * ldb + b] : 0.f) + sum / k;
  }
}

// ---------------------------------------------------------------- temporal pooling
// vf = h + (u ? u : 0); out[b] = (1/F) sum_f vf/||vf||;  norms[b][f] saved.  One block per video.
__global__ __launch_bounds__(256) void temporal_pool_fwd_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                                float* __restrict__ out, float* __restrict__ norms,
                                                                int F, int D) {
  __shared__ float red[4][1024];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = lane; c < D; c += 64) red[w][c] = 0.f;
  for (int f = w; f < F; f += 4) {
    const long off = ((long)b * F + f) * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
      float v = h[off + c] + (u ? u[off + c] : 0.f);
      s += v * v;
    }
    float n = sqrtf(wave_sum(s));
    if (lane == 0) norms[(long)b * F + f] = n;
    float inv = 1.0f / n;
    for (int c = lane; c < D; c += 64) red[w][c] += (h[off + c] + (u ? u[off + c] : 0.f)) * inv;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) out[(long)b * D + c] = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / F;
}

// dvf[b][f] = (1/F) * (dout[b] - yn * <yn, dout[b]>) / n, yn = vf / n
__global__ __launch_bounds__(256) void temporal_pool_bwd_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                                const float* __restrict__ norms,
                                                                const float* __restrict__ dout, float* __restrict__ dvf,
                                                                int rows, int F, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / F;
  const long off = (long)row * D;
  const float inv = 1.0f / norms[row];
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += (h[off + c] + (u ? u[off + c] : 0.f)) * inv * dout[(long)b * D + c];
  s = wave_sum(s);
  for (int c = lane; c < D; c += 64) {
    float yn = (h[off + c] + (u ? u[off + c] : 0.f)) * inv;
    dvf[off + c] = (dout[(long)b * D + c] - yn * s) * inv / F;
  }
}

// out[r][:] = x[r][:] + table[r % period][:]
__global__ __launch_bounds__(256) void add_rowbias_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                          float* __restrict__ out, long rows, int period, int D) {
  const long total = rows * (D / 4);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx / (D / 4);
    int c = (int)(idx % (D / 4)) * 4;
    f4 a = *reinterpret_cast<const f4*>(x + r * D + c);
    f4 t = *reinterpret_cast<const f4*>(table + (long)(r % period) * D + c);
    *reinterpret_cast<f4*>(out + r * D + c) = a + t;
  }
}

// ---------------------------------------------------------------- temporal attention, fp32, F <= 64, head dim 64
// One wave per (video, head).  qkv: [b*F, 3*D] (Q | K | V); out: [b*F, D]; probs saved [b, H, F, F].
__global__ __launch_bounds__(64) void tattn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ probs, int F, int H, int causal) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x;
  const int D = H * 64;
  float* q = sm; float* k = q + F * 65; float* v = k + F * 65; float* p = v + F * 65;   // p: [F][F]
  for (int idx = lane; idx < F * 64; idx += 64) {
    int f = idx >> 6, d = idx & 63;
    const float* base = qkv + ((long)b * F + f) * 3 * D + hd * 64 + d;
    q[f * 65 + d] = base[0] * 0.125f;
    k[f * 65 + d] = base[D];
    v[f * 65 + d] = base[2 * D];
  }
  __syncthreads();
  for (int idx = lane; idx < F * F; idx += 64) {
    int i = idx / F, j = idx % F;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) s += q[i * 65 + d] * k[j * 65 + d];
    p[idx] = (causal && j > i) ? -INFINITY : s;
  }
  __syncthreads();
  if (lane < F) {
    float m = -INFINITY;
    for (int j = 0; j < F; ++j) m = fmaxf(m, p[lane * F + j]);
    float s = 0.f;
    for (int j = 0; j < F; ++j) { float e = __expf(p[lane * F + j] - m); p[lane * F + j] = e; s += e; }
    float inv = 1.0f / s;
    for (int j = 0; j < F; ++j) p[lane * F + j] *= inv;
  }
  __syncthreads();
  float* pg = probs + ((long)b * H + hd) * F * F;
  for (int idx = lane; idx < F * F; idx += 64) pg[idx] = p[idx];
  for (int i = 0; i < F; ++i) {
    float o = 0.f;
    for (int j = 0; j < F; ++j) o += p[i * F + j] * v[j * 65 + lane];
    out[((long)b * F + i) * D + hd * 64 + lane] = o;
  }
}

__global__ __launch_bounds__(64) void tattn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                       const float* __restrict__ dout, float* __restrict__ dqkv, int F,
                                                       int H) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x;
  const int D = H * 64;
  float* q = sm; float* k = q + F * 65; float* v = k + F * 65; float* dO = v + F * 65;
  float* p = dO + F * 65; float* ds = p + F * F;
  for (int idx = lane; idx < F * 64; idx += 64) {
    int f = idx >> 6, d = idx & 63;
    const float* base = qkv + ((long)b * F + f) * 3 * D + hd * 64 + d;
    q[f * 65 + d] = base[0];
    k[f * 65 + d] = base[D];
    v[f * 65 + d] = base[2 * D];
    dO[f * 65 + d] = dout[((long)b * F + f) * D + hd * 64 + d];
  }
  const float* pg = probs + ((long)b * H + hd) * F * F;
  for (int idx = lane; idx < F * F; idx += 64) p[idx] = pg[idx];
  __syncthreads();
  // dP[i][j] = <dO[i], V[j]>
  for (int idx = lane; idx < F * F; idx += 64) {
    int i = idx / F, j = idx % F;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) s += dO[i * 65 + d] * v[j * 65 + d];
    ds[idx] = s;
  }
  __syncthreads();
  if (lane < F) {
    float dot = 0.f;
    for (int j = 0; j < F; ++j) dot += p[lane * F + j] * ds[lane * F + j];
    for (int j = 0; j < F; ++j) ds[lane * F + j] = p[lane * F + j] * (ds[lane * F + j] - dot) * 0.125f;
  }
  __syncthreads();
  for (int i = 0; i < F; ++i) {
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < F; ++j) {
      dq += ds[i * F + j] * k[j * 65 + lane];
      dk += ds[j * F + i] * q[j * 65 + lane];
      dv += p[j * F + i] * dO[j * 65 + lane];
    }
    float* o = dqkv + ((long)b * F + i) * 3 * D + hd * 64 + lane;
    o[0] = dq; o[D] = dk; o[2 * D] = dv;
  }
}

// ---------------------------------------------------------------- fp32 attention for 65..256 tokens (the CLIP towers in the
// reference's fp32-upcast regime, model.float(): ViT-B/16 has 197 tokens, the text tower up to 77).  Exact fp32 arithmetic,
// probabilities saved like the short kernel's; a parity path, not a fast one.  One 256-thread workgroup per (sequence, head):
// K and V in LDS ([L][65] floats: a lane per key row or per feature, both conflict-free), a wave per query row.
constexpr int GA_MAXL = 256;
__global__ __launch_bounds__(256) void gattn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        float* __restrict__ probs, int L, int H, int causal) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = H * 64;
  float* k = sm; float* v = k + L * 65; float* prow = v + L * 65;            // prow: [4][GA_MAXL], qrow: [4][64]
  float* qrow = prow + 4 * GA_MAXL;
  for (int idx = threadIdx.x; idx < L * 64; idx += 256) {
    const int f = idx >> 6, d = idx & 63;
    const float* base = qkv + ((long)b * L + f) * 3 * D + hd * 64 + d;
    k[f * 65 + d] = base[D];
    v[f * 65 + d] = base[2 * D];
  }
  __syncthreads();
  float* pw = prow + w * GA_MAXL;
  float* qw = qrow + w * 64;
  float* pg = probs + ((long)b * H + hd) * L * L;
  for (int i = w; i < L; i += 4) {
    qw[lane] = qkv[((long)b * L + i) * 3 * D + hd * 64 + lane] * 0.125f;
    float sc[4];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      float s = -INFINITY;
      if (j < L && !(causal && j > i)) {
        s = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) s += qw[d] * k[j * 65 + d];
      }
      sc[t] = s;
      m = fmaxf(m, s);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) { sc[t] = (sc[t] == -INFINITY) ? 0.f : __expf(sc[t] - m); sum += sc[t]; }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      if (j < L) { const float pj = sc[t] * inv; pw[j] = pj; pg[(long)i * L + j] = pj; }
    }
    float o = 0.f;
    for (int j = 0; j < L; ++j) o += pw[j] * v[j * 65 + lane];
    out[((long)b * L + i) * D + hd * 64 + lane] = o;
  }
}

// dV[j] = sum_i P[i][j] dO[i];  dP[i][j] = <dO[i], V[j]>;  dS = P o (dP - rowsum(P o dP));  dQ[i] = sum_j dS[i][j] K[j] / 8;
// dK[j] = sum_i dS[i][j] Q[i] / 8.  Pass A: a wave per query row (dQ, and delta[i] = rowsum(P o dP) into LDS); pass B: a wave
// per key row (dK, dV) re-forming dP[i][j] by a wave reduction.  No atomics: every output element has one owner.
__global__ __launch_bounds__(256) void gattn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                        const float* __restrict__ dout, float* __restrict__ dqkv, int L, int H) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int D = H * 64;
  float* k = sm; float* v = k + L * 65; float* drow = v + L * 65;            // drow: [4][GA_MAXL] dS of a row; delta: [GA_MAXL]
  float* delta = drow + 4 * GA_MAXL;
  for (int idx = threadIdx.x; idx < L * 64; idx += 256) {
    const int f = idx >> 6, d = idx & 63;
    const float* base = qkv + ((long)b * L + f) * 3 * D + hd * 64 + d;
    k[f * 65 + d] = base[D];
    v[f * 65 + d] = base[2 * D];
  }
  __syncthreads();
  const float* pg = probs + ((long)b * H + hd) * L * L;
  float* dw = drow + w * GA_MAXL;
  for (int i = w; i < L; i += 4) {
    const float dO = dout[((long)b * L + i) * D + hd * 64 + lane];
    float dot = 0.f;
    for (int j = 0; j < L; ++j) {
      const float dp = wave_sum(dO * v[j * 65 + lane]);
      if (lane == 0) dw[j] = dp;
      dot += pg[(long)i * L + j] * dp;
    }
    if (lane == 0) delta[i] = dot;
    float dq = 0.f;
    for (int j = 0; j < L; ++j) dq += pg[(long)i * L + j] * (dw[j] - dot) * k[j * 65 + lane];
    dqkv[((long)b * L + i) * 3 * D + hd * 64 + lane] = dq * 0.125f;
  }
  __syncthreads();
  for (int j = w; j < L; j += 4) {
    const float vj = v[j * 65 + lane];
    float dk = 0.f, dv = 0.f;
    for (int i = 0; i < L; ++i) {
      const float dO = dout[((long)b * L + i) * D + hd * 64 + lane];
      const float q = qkv[((long)b * L + i) * 3 * D + hd * 64 + lane];
      const float pij = pg[(long)i * L + j];
      const float dp = wave_sum(dO * vj);
      dv += pij * dO;
      dk += pij * (dp - delta[i]) * q;
    }
    float* o = dqkv + ((long)b * L + j) * 3 * D + hd * 64 + lane;
    o[D] = dk * 0.125f;
    o[2 * D] = dv;
  }
}

inline int nblk(long work, int cap = 4096) {
  long b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int hmmc_l2norm_fwd(const float* x, float* y, float* norm, int rows, int D, float eps, hipStream_t stream) {
  if (!x || !y || !norm || rows <= 0 || (D & 3)) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, norm, rows, D, eps);
  return hmmc_launch_status();
}

extern "C" int hmmc_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int rows, int D,
                               hipStream_t stream) {
  if (!dy || !y || !norm || !dx || rows <= 0 || (D & 3)) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dy, y, norm, dx, rows, D);
  return hmmc_launch_status();
}

extern "C" int hmmc_infonce_fwd(const float* S, float* lse_row, float* lse_col, float* loss, int B, int F, float w_video,
                                float w_frame, hipStream_t stream) {
  if (!S || !lse_row || !lse_col || !loss || B <= 0 || F < 0) return HMMC_ERR_ARG;
  long C = (long)B * (1 + F);
  hipLaunchKernelGGL(infonce_rowlse_kernel, dim3(B, 1 + F), dim3(64), 0, stream, S, lse_row, B, F);
  hipLaunchKernelGGL(infonce_collse_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, stream, S, lse_col, B, C);
  hipLaunchKernelGGL(infonce_loss_kernel, dim3(1), dim3(256), 0, stream, S, (const float*)lse_row, (const float*)lse_col,
                     loss, B, F, w_video, w_frame);
  return hmmc_launch_status();
}

extern "C" int hmmc_infonce_bwd(const float* S, const float* lse_row, const float* lse_col, const float* grad_out,
                                float* dS, int B, int F, float w_video, float w_frame, hipStream_t stream) {
  if (!S || !lse_row || !lse_col || !grad_out || !dS || B <= 0 || F < 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3(nblk((long)B * B * (1 + F))), dim3(256), 0, stream, S, lse_row, lse_col,
                     grad_out, dS, B, F, w_video, w_frame);
  return hmmc_launch_status();
}

// rank[q] = number of candidates scoring strictly higher than query q's target (metrics.py:12-20: position of the
// diagonal in the descending sort, first position on ties).  One wave per query; transposed != 0 ranks the columns of S
// (video -> text) without materialising S^T.
__global__ __launch_bounds__(256) void retrieval_rank_kernel(const float* __restrict__ S, const int* __restrict__ target,
                                                             int* __restrict__ rank, int Q, int V, long ld, int transposed) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  const int t = target ? target[q] : q;
  const float ref = transposed ? S[(long)t * ld + q] : S[(long)q * ld + t];
  float cnt = 0.f;
  for (int j = lane; j < V; j += 64) {
    float v = transposed ? S[(long)j * ld + q] : S[(long)q * ld + j];
    cnt += v > ref ? 1.f : 0.f;
  }
  cnt = wave_sum(cnt);
  if (lane == 0) rank[q] = (int)cnt;
}

// out[g][v] = max over the sentences s of group g (offsets[g] <= s < offsets[g + 1]) of S[s][v]: the video-to-text matrix of
// multi-sentence retrieval (metrics.py:79-86 tensor_video_to_text_sim: max over a video's captions; NaN counts as -inf),
// stored [groups][videos] so that hmmc_retrieval_rank(transposed = 1) ranks it without a transposed copy.
__global__ __launch_bounds__(256) void segment_max_kernel(const float* __restrict__ S, const int* __restrict__ offsets,
                                                          float* __restrict__ out, int G, int V, long ld) {
  const int g = blockIdx.y;
  const int s0 = offsets[g], s1 = offsets[g + 1];
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
    float m = -INFINITY;
    for (int s = s0; s < s1; ++s) m = fmaxf(m, S[(long)s * ld + v]);
    out[(long)g * V + v] = m;
  }
}

extern "C" int hmmc_segment_max(const float* S, const int* offsets, float* out, int groups, int V, long ld, hipStream_t stream) {
  if (!S || !offsets || !out || groups <= 0 || V <= 0 || ld < V) return HMMC_ERR_ARG;
  if (groups > 65535) return HMMC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(segment_max_kernel, dim3((V + 255) / 256, groups), dim3(256), 0, stream, S, offsets, out, groups, V, ld);
  return hmmc_launch_status();
}

extern "C" int hmmc_topk_mean(const float* S_frame, const float* base, float* out, int bq, int bv, int F, int k, long lds,
                              long ldb, hipStream_t stream) {
  if (!S_frame || !out || bq <= 0 || bv <= 0 || F <= 0 || F > 64 || k <= 0 || k > F) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(topk_mean_kernel, dim3(nblk((long)bq * bv)), dim3(256), 0, stream, S_frame, base, out, bq, bv, F, k, lds,
                     ldb);
  return hmmc_launch_status();
}

extern "C" int hmmc_retrieval_rank(const float* S, const int* target, int* rank, int Q, int V, long ld, int transposed,
                                   hipStream_t stream) {
  if (!S || !rank || Q <= 0 || V <= 0 || ld <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(retrieval_rank_kernel, dim3((Q + 3) / 4), dim3(256), 0, stream, S, target, rank, Q, V, ld, transposed);
  return hmmc_launch_status();
}

extern "C" int hmmc_temporal_pool_fwd(const float* h, const float* u, float* out, float* norms, int b, int F, int D,
                                      hipStream_t stream) {
  if (!h || !out || !norms || b <= 0 || F <= 0 || D > 1024) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(temporal_pool_fwd_kernel, dim3(b), dim3(256), 0, stream, h, u, out, norms, F, D);
  return hmmc_launch_status();
}

extern "C" int hmmc_temporal_pool_bwd(const float* h, const float* u, const float* norms, const float* dout, float* dvf,
                                      int b, int F, int D, hipStream_t stream) {
  if (!h || !norms || !dout || !dvf || b <= 0 || F <= 0) return HMMC_ERR_ARG;
  int rows = b * F;
  hipLaunchKernelGGL(temporal_pool_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, h, u, norms, dout, dvf, rows, F, D);
  return hmmc_launch_status();
}

extern "C" int hmmc_add_rowbias(const float* x, const float* table, float* out, long rows, int period, int D,
                                hipStream_t stream) {
  if (!x || !table || !out || rows <= 0 || period <= 0 || (D & 3)) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(add_rowbias_kernel, dim3(nblk(rows * (D / 4))), dim3(256), 0, stream, x, table, out, rows, period, D);
  return hmmc_launch_status();
}

extern "C" int hmmc_temporal_attention_fwd(const float* qkv, float* out, float* probs, int b, int F, int H, int causal,
                                           hipStream_t stream) {
  if (!qkv || !out || !probs || b <= 0 || F <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (F > GA_MAXL) return HMMC_ERR_UNSUPPORTED;
  if (F > 64) {                                  // the CLIP towers' token counts in the fp32 regime (parity path)
    size_t gl = (size_t)(2 * F * 65 + 4 * GA_MAXL + 4 * 64) * sizeof(float);
    static bool gdone[HMMC_MAX_DEVICES] = {false};
    hmmc_allow_lds((const void*)gattn_fwd_kernel, 160 * 1024 - 4096, gdone);
    hipLaunchKernelGGL(gattn_fwd_kernel, dim3(b * H), dim3(256), gl, stream, qkv, out, probs, F, H, causal);
    return hmmc_launch_status();
  }
  size_t lds = (size_t)(3 * F * 65 + F * F) * sizeof(float);
  static bool done[HMMC_MAX_DEVICES] = {false};
  hmmc_allow_lds((const void*)tattn_fwd_kernel, 160 * 1024 - 4096, done);
  hipLaunchKernelGGL(tattn_fwd_kernel, dim3(b * H), dim3(64), lds, stream, qkv, out, probs, F, H, causal);
  return hmmc_launch_status();
}

extern "C" int hmmc_temporal_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int b,
                                           int F, int H, hipStream_t stream) {
  if (!qkv || !probs || !dout || !dqkv || b <= 0 || F <= 0 || H <= 0) return HMMC_ERR_ARG;
  if (F > GA_MAXL) return HMMC_ERR_UNSUPPORTED;
  if (F > 64) {
    size_t gl = (size_t)(2 * F * 65 + 4 * GA_MAXL + GA_MAXL) * sizeof(float);
    static bool gdone[HMMC_MAX_DEVICES] = {false};
    hmmc_allow_lds((const void*)gattn_bwd_kernel, 160 * 1024 - 4096, gdone);
    hipLaunchKernelGGL(gattn_bwd_kernel, dim3(b * H), dim3(256), gl, stream, qkv, probs, dout, dqkv, F, H);
    return hmmc_launch_status();
  }
  size_t lds = (size_t)(4 * F * 65 + 2 * F * F) * sizeof(float);
  static bool done[HMMC_MAX_DEVICES] = {false};
  hmmc_allow_lds((const void*)tattn_bwd_kernel, 160 * 1024 - 4096, done);
  hipLaunchKernelGGL(tattn_bwd_kernel, dim3(b * H), dim3(64), lds, stream, qkv, probs, dout, dqkv, F, H);
  return hmmc_launch_status();
}
