// Multi-tensor optimizer-side kernels of the HMMC hot path (HBM-bound):
//   * BertAdam.step            (reference modules/optimization.py:103-168)   K16 of SURVEY.md section 2.3
//   * global clip_grad_norm_   (main_task_retrieval.py:291)
//   * momentum-encoder EMA     (modules/modeling.py:238-242)                  K12
//   * queue enqueue            (modules/modeling.py:244-284)                  K13
// The reference loops over 350-560 tensors in Python with ~10 launches each; here one launch covers
// every tensor through a device-side table.  Arithmetic is evaluated op by op in each tensor's own
// dtype with the reference's rounding points (fp16 params keep fp16 moments and round after every
// tensor op), so fp16 results are bit-identical to the reference expression.
//
// tab  : int64 [T][8]  = { p, g, m, v (device pointers), numel, dtype (0 fp16 / 1 fp32), group, 0 }
// groups (by value, kernel argument): float [G][8] = { lr_scheduled, weight_decay, b1, b2, eps, max_grad_norm, 1-b1, 1-b2 }
//        — the per-step scalars travel in the kernel argument buffer: no per-step host-to-device copy, no sync
// chunk: int32 [C][2]  = { tensor index, chunk index }   (CHUNK elements per block)
#include "common.h"

namespace {

constexpr int CHUNK = 32768;

// The reference evaluates every fp16 tensor op as: fp32 arithmetic, round to fp32, then round to fp16.
// hipcc would fuse "multiply, convert" into v_fma_mixlo_f16 (ONE rounding from the exact product) and
// contract separate mul/add into fma; both differ from the reference in the last fp16 bit on ties.
// opq() / r16s() (common.h) make a value opaque to the optimizer, pinning an fp32 rounding point.

__device__ __forceinline__ float block_sum(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// part[chunk] = sum of g^2 over this chunk (16-byte loads; the scalar loop only handles a tensor's ragged tail).  No
// atomics: mt_sumsq_finish_kernel adds a tensor's chunks in a fixed order, so norms - and with them the clip coefficient
// and every updated weight - are bit-identical from run to run.
__global__ __launch_bounds__(256) void mt_sumsq_kernel(const long* __restrict__ tab, const int* __restrict__ chunk,
                                                       float* __restrict__ part) {
  const int t = chunk[2 * blockIdx.x], ci = chunk[2 * blockIdx.x + 1];
  const long* e = tab + (long)t * 8;
  const long n = e[4];
  const long i0 = (long)ci * CHUNK, i1 = min(n, i0 + CHUNK);
  float s = 0.f;
  if (e[5] == 0) {
    const half_t* g = reinterpret_cast<const half_t*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~7L);
    for (long i = i0 + threadIdx.x * 8; i < iv; i += 256 * 8) {
      h8 v = *reinterpret_cast<const h8*>(g + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { float x = (float)v[j]; s += x * x; }
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) { float x = (float)g[i]; s += x * x; }
  } else {
    const float* g = reinterpret_cast<const float*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~3L);
    for (long i = i0 + threadIdx.x * 4; i < iv; i += 256 * 4) {
      f4 v = *reinterpret_cast<const f4*>(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) { float x = g[i]; s += x * x; }
  }
  s = block_sum(s);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// Squared norm of tensor t = sum of the partial sums of its chunks: chunks tab[t][7] .. + ceil(numel / CHUNK) - 1 of the list,
// lane l takes chunks l, l + 64, ... in order, then a fixed butterfly: ONE wave, every lane returns the total.  This is the only
// place a tensor's partials are added, so every kernel that needs the norm (the finish kernel, the clip coefficient, BertAdam's
// per-parameter clip) gets the same bits.
__device__ __forceinline__ float tensor_sumsq(const long* __restrict__ tab, const int* __restrict__ chunk, int nchunks,
                                              const float* __restrict__ part, int t, int lane) {
  const long* e = tab + (long)t * 8;
  const long first = e[7];
  const long nch = (e[4] + CHUNK - 1) / CHUNK;
  if (first < 0 || first + nch > nchunks || (nch > 0 && chunk[2 * first] != t)) return __builtin_nanf("");   // table without the first-chunk column
  float s = 0.f;
  for (long c = lane; c < nch; c += 64) s += part[first + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

// sumsq[t] for every tensor.  One wave per tensor.
__global__ __launch_bounds__(64) void mt_sumsq_finish_kernel(const long* __restrict__ tab, const int* __restrict__ chunk,
                                                             int nchunks, const float* __restrict__ part, float* __restrict__ sumsq) {
  const float s = tensor_sumsq(tab, chunk, nchunks, part, blockIdx.x, threadIdx.x);
  if (threadIdx.x == 0) sumsq[blockIdx.x] = s;
}

constexpr int FINISH_COEF_MAX_T = 4096;
// mt_sumsq_finish_kernel and mt_clip_coef_kernel in ONE single-block launch (16 waves share the tensors): sumsq[t] for every
// tensor, then out[0] = clip coefficient, out[1] = total norm - the same sums in the same order as the two kernels.
__global__ __launch_bounds__(1024) void mt_finish_coef_kernel(const long* __restrict__ tab, const int* __restrict__ chunk, int nchunks,
                                                              const float* __restrict__ part, float* __restrict__ sumsq, int T,
                                                              float max_norm, float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ float ssq[FINISH_COEF_MAX_T];           // the norms travel to the second half through LDS, not through the global array
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += 16) {
    const float s = tensor_sumsq(tab, chunk, nchunks, part, t, lane);
    if (lane == 0) { sumsq[t] = s; ssq[t] = s; }
  }
  __syncthreads();
  if (threadIdx.x < 256) {                          // exactly mt_clip_coef_kernel's arithmetic: 256 threads, block_sum's order
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
      float n = sqrtf(ssq[t]);
      if (tab[(long)t * 8 + 5] == 0) n = r16(n);
      s += n * n;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float c = max_norm / (total + 1e-6f);
    out[0] = c < 1.0f ? c : 1.0f;
    out[1] = total;
  }
}

// out[0] = clip coefficient, out[1] = total norm (torch.nn.utils.clip_grad_norm_ semantics: per-tensor
// norms are rounded to the tensor's dtype, the stack of norms is fp32)
__global__ __launch_bounds__(256) void mt_clip_coef_kernel(const long* __restrict__ tab, const float* __restrict__ sumsq,
                                                           int T, float max_norm, float* __restrict__ out) {
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    float n = sqrtf(sumsq[t]);
    if (tab[(long)t * 8 + 5] == 0) n = r16(n);
    s += n * n;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) {
    float total = sqrtf(s);
    float c = max_norm / (total + 1e-6f);
    out[0] = c < 1.0f ? c : 1.0f;
    out[1] = total;
  }
}

// part_after (optional): [nchunks] partial sums of squares of the gradient AS WRITTEN (scaled and rounded), accumulated in
// mt_sumsq_kernel's order, so that the optimizer's per-parameter clip can take its norms from this pass instead of reading
// every gradient again (hmmc_mt_clip_grad_norm_keep / hmmc_mt_bertadam_ext); with a coefficient of 1 nothing is written to the
// gradients and the partial sums of the first pass are handed on.
__global__ __launch_bounds__(256) void mt_scale_kernel(const long* __restrict__ tab, const int* __restrict__ chunk,
                                                       const float* __restrict__ coef, const float* __restrict__ part_before,
                                                       float* __restrict__ part_after) {
  const float c = coef[0];
  if (c == 1.0f) {
    if (part_after && threadIdx.x == 0) part_after[blockIdx.x] = part_before[blockIdx.x];
    return;
  }
  const int t = chunk[2 * blockIdx.x], ci = chunk[2 * blockIdx.x + 1];
  const long* e = tab + (long)t * 8;
  const long n = e[4];
  const long i0 = (long)ci * CHUNK, i1 = min(n, i0 + CHUNK);
  float s = 0.f;
  if (e[5] == 0) {
    half_t* g = reinterpret_cast<half_t*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~7L);
    for (long i = i0 + threadIdx.x * 8; i < iv; i += 256 * 8) {
      h8 v = *reinterpret_cast<h8*>(g + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = (half_t)opq((float)v[j] * c); float x = (float)v[j]; s += x * x; }
      *reinterpret_cast<h8*>(g + i) = v;
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) { g[i] = (half_t)opq((float)g[i] * c); float x = (float)g[i]; s += x * x; }
  } else {
    float* g = reinterpret_cast<float*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~3L);
    for (long i = i0 + threadIdx.x * 4; i < iv; i += 256 * 4) {
      f4 v = *reinterpret_cast<f4*>(g + i);
      v = v * c;
      *reinterpret_cast<f4*>(g + i) = v;
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) { g[i] = g[i] * c; float x = g[i]; s += x * x; }
  }
  if (part_after) {                                   // kernel-argument uniform
    s = block_sum(s);
    if (threadIdx.x == 0) part_after[blockIdx.x] = s;
  }
}

// BertAdam.step for every tensor.  sumsq[t] = sum g^2 of the gradient as it stands (per-parameter clip).
constexpr int MAX_GROUPS = 32;
struct AdamGroups { float v[MAX_GROUPS][8]; };

// index (optional): sumsq is another table's array and tensor t's entry is sumsq[index[t]] (hmmc_mt_bertadam_ext)
// The per-parameter clip multiplies by c = min(1, max_grad_norm / (norm + 1e-6)); where c is exactly 1 - every tensor whose norm
// is below max_grad_norm, i.e. every tensor right after a global clip to the same bound - g * 1 is g and the gradient is not
// written back (the reference's in-place clip leaves the same bits): 6 instead of 7 passes over the parameter bytes.
__global__ __launch_bounds__(256) void mt_bertadam_kernel(const long* __restrict__ tab, AdamGroups groups,
                                                          const int* __restrict__ chunk, const float* __restrict__ sumsq_,
                                                          const int* __restrict__ index) {
  const int t = chunk[2 * blockIdx.x], ci = chunk[2 * blockIdx.x + 1];
  const float* const sumsq = index ? sumsq_ + index[t] - t : sumsq_;       // sumsq[t] below = the tensor's squared norm
  const long* e = tab + (long)t * 8;
  const float* f = groups.v[e[6]];
  const long n = e[4];
  const long i0 = (long)ci * CHUNK, i1 = min(n, i0 + CHUNK);
  const float lr = f[0], wd = f[1], b1 = f[2], b2 = f[3], eps = f[4], maxn = f[5];
  const float ob1 = f[6], ob2 = f[7];     // 1-b1, 1-b2 evaluated in double on the host, as the reference does
  if (e[5] == 0) {
    half_t* p = reinterpret_cast<half_t*>(e[0]);
    half_t* g = reinterpret_cast<half_t*>(e[1]);
    half_t* m = reinterpret_cast<half_t*>(e[2]);
    half_t* v = reinterpret_cast<half_t*>(e[3]);
    // Tensor.add_(g, alpha=1-b1) on an fp16 tensor rounds alpha itself to fp16 on the reference's CPU
    // path (every other scalar stays fp32); the golden vectors pin this.
    const float ob1h = r16(ob1);
    float c = 1.0f;
    if (maxn > 0.f) {
      float nrm = r16(sqrtf(sumsq[t]));
      c = r16(maxn / r16(nrm + 1e-6f));
      c = c < 1.0f ? c : 1.0f;
    }
    const bool wr_g = maxn > 0.f && c != 1.0f;     // fp16(g * 1) == g
    auto upd = [&](half_t& pp, half_t& gg, half_t& mm, half_t& vv) {
      float gi = (float)gg;
      if (wr_g) { gi = r16s(gi * c); gg = (half_t)gi; }
      // add_(g, alpha) is a true fma on the reference's CPU path; addcmul_ is (value*g)*g then an add
      float mi = r16s(__fmaf_rn(ob1h, gi, r16s((float)mm * b1)));
      float vi = r16s(r16s((float)vv * b2) + opq(opq(ob2 * gi) * gi));
      float pi = (float)pp;
      float u = r16s(mi / r16s(r16s(sqrtf(vi)) + eps));
      if (wd > 0.f) u = r16s(u + r16s(wd * pi));
      float uw = r16s(lr * u);
      pp = (half_t)opq(pi - uw);
      mm = (half_t)mi;
      vv = (half_t)vi;
    };
    const long iv = i0 + ((i1 - i0) & ~7L);
    for (long i = i0 + threadIdx.x * 8; i < iv; i += 256 * 8) {
      h8 P = *reinterpret_cast<h8*>(p + i), G = *reinterpret_cast<h8*>(g + i);
      h8 Mv = *reinterpret_cast<h8*>(m + i), Vv = *reinterpret_cast<h8*>(v + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { half_t a = P[j], b = G[j], cc = Mv[j], d = Vv[j]; upd(a, b, cc, d); P[j] = a; G[j] = b; Mv[j] = cc; Vv[j] = d; }
      *reinterpret_cast<h8*>(p + i) = P;
      if (wr_g) *reinterpret_cast<h8*>(g + i) = G;
      *reinterpret_cast<h8*>(m + i) = Mv;
      *reinterpret_cast<h8*>(v + i) = Vv;
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) upd(p[i], g[i], m[i], v[i]);
  } else {
    float* p = reinterpret_cast<float*>(e[0]);
    float* g = reinterpret_cast<float*>(e[1]);
    float* m = reinterpret_cast<float*>(e[2]);
    float* v = reinterpret_cast<float*>(e[3]);
    float c = 1.0f;
    if (maxn > 0.f) {
      c = maxn / (sqrtf(sumsq[t]) + 1e-6f);
      c = c < 1.0f ? c : 1.0f;
    }
    const bool wr_g = maxn > 0.f && c != 1.0f;     // g * 1.0f == g
    auto upd = [&](float& pp, float& gg, float& mm, float& vv) {
      float gi = gg;
      if (wr_g) { gi = gi * c; gg = gi; }
      float mi = __fmaf_rn(ob1, gi, opq(mm * b1));
      float vi = opq(vv * b2) + opq(opq(ob2 * gi) * gi);
      float u = mi / opq(sqrtf(vi) + eps);
      if (wd > 0.f) u = opq(u) + opq(wd * pp);
      pp = pp - opq(lr * opq(u));
      mm = mi;
      vv = vi;
    };
    const long iv = i0 + ((i1 - i0) & ~3L);
    for (long i = i0 + threadIdx.x * 4; i < iv; i += 256 * 4) {
      f4 P = *reinterpret_cast<f4*>(p + i), G = *reinterpret_cast<f4*>(g + i);
      f4 Mv = *reinterpret_cast<f4*>(m + i), Vv = *reinterpret_cast<f4*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) { float a = P[j], b = G[j], cc = Mv[j], d = Vv[j]; upd(a, b, cc, d); P[j] = a; G[j] = b; Mv[j] = cc; Vv[j] = d; }
      *reinterpret_cast<f4*>(p + i) = P;
      if (wr_g) *reinterpret_cast<f4*>(g + i) = G;
      *reinterpret_cast<f4*>(m + i) = Mv;
      *reinterpret_cast<f4*>(v + i) = Vv;
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) upd(p[i], g[i], m[i], v[i]);
  }
}

// EMA: tab rows = { p_k, p, 0, 0, numel, dtype }: p_k = r(r(p_k * mom) + r(p * (1 - mom)))
__global__ __launch_bounds__(256) void mt_ema_kernel(const long* __restrict__ tab, const int* __restrict__ chunk,
                                                     float mom, float one_minus) {
  const int t = chunk[2 * blockIdx.x], ci = chunk[2 * blockIdx.x + 1];
  const long* e = tab + (long)t * 8;
  const long n = e[4];
  const long i0 = (long)ci * CHUNK, i1 = min(n, i0 + CHUNK);
  if (e[5] == 0) {
    half_t* pk = reinterpret_cast<half_t*>(e[0]);
    const half_t* p = reinterpret_cast<const half_t*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~7L);
    for (long i = i0 + threadIdx.x * 8; i < iv; i += 256 * 8) {
      h8 a = *reinterpret_cast<h8*>(pk + i), b = *reinterpret_cast<const h8*>(p + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = (half_t)opq(r16s((float)a[j] * mom) + r16s((float)b[j] * one_minus));
      *reinterpret_cast<h8*>(pk + i) = a;
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256)
      pk[i] = (half_t)opq(r16s((float)pk[i] * mom) + r16s((float)p[i] * one_minus));
  } else {
    float* pk = reinterpret_cast<float*>(e[0]);
    const float* p = reinterpret_cast<const float*>(e[1]);
    const long iv = i0 + ((i1 - i0) & ~3L);
    for (long i = i0 + threadIdx.x * 4; i < iv; i += 256 * 4) {
      f4 a = *reinterpret_cast<f4*>(pk + i), b = *reinterpret_cast<const f4*>(p + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) { float x = opq(a[j] * mom), y = opq(b[j] * one_minus); a[j] = x + y; }   // three tensor ops, no fma
      *reinterpret_cast<f4*>(pk + i) = a;
    }
    for (long i = iv + threadIdx.x; i < i1; i += 256) {
      float a = opq(pk[i] * mom), b = opq(p[i] * one_minus);
      pk[i] = a + b;
    }
  }
}

// queue[d][col0 + r] = keys[r][d] / max(||keys[r]||, 1e-12);  queue: [E][W]
__global__ __launch_bounds__(256) void enqueue_kernel(const float* __restrict__ keys, float* __restrict__ queue, int R,
                                                      int E, long W, long col0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* kr = keys + (long)r * E;
  float s = 0.f;
  for (int d = lane; d < E; d += 64) s += kr[d] * kr[d];
  float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int d = lane; d < E; d += 64) queue[(long)d * W + col0 + r] = kr[d] * inv;
}

}  // namespace

extern "C" int hmmc_mt_sumsq(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, hipStream_t stream) {
  if (!tab || !chunk || !sumsq || nchunks <= 0 || T <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, sumsq + T);
  hipLaunchKernelGGL(mt_sumsq_finish_kernel, dim3(T), dim3(64), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq);
  return hmmc_launch_status();
}

// grads *= min(1, max_norm / (total_norm + 1e-6)); out[0] = coefficient, out[1] = total_norm
extern "C" int hmmc_mt_clip_grad_norm(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, float max_norm,
                                      float* out, hipStream_t stream) {
  if (!tab || !chunk || !sumsq || !out || nchunks <= 0 || T <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, sumsq + T);
  if (T <= FINISH_COEF_MAX_T) {
    hipLaunchKernelGGL(mt_finish_coef_kernel, dim3(1), dim3(1024), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq, T, max_norm, out);
  } else {
    hipLaunchKernelGGL(mt_sumsq_finish_kernel, dim3(T), dim3(64), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq);
    hipLaunchKernelGGL(mt_clip_coef_kernel, dim3(1), dim3(256), 0, stream, tab, (const float*)sumsq, T, max_norm, out);
  }
  hipLaunchKernelGGL(mt_scale_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, (const float*)out, (const float*)nullptr, (float*)nullptr);
  return hmmc_launch_status();
}

// The same, and sumsq_after[0 .. T) = the squared norm of every gradient as it stands AFTER the clip (sumsq_after: float
// [T + nchunks], laid out like sumsq), formed inside the scaling pass: what BertAdam's per-parameter clip needs next
// (hmmc_mt_bertadam_ext), without another pass over the gradients.
extern "C" int hmmc_mt_clip_grad_norm_keep(const long* tab, const int* chunk, int nchunks, float* sumsq, int T, float max_norm,
                                           float* out, float* sumsq_after, hipStream_t stream) {
  if (!tab || !chunk || !sumsq || !out || !sumsq_after || nchunks <= 0 || T <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, sumsq + T);
  if (T <= FINISH_COEF_MAX_T) {
    hipLaunchKernelGGL(mt_finish_coef_kernel, dim3(1), dim3(1024), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq, T, max_norm, out);
  } else {
    hipLaunchKernelGGL(mt_sumsq_finish_kernel, dim3(T), dim3(64), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq);
    hipLaunchKernelGGL(mt_clip_coef_kernel, dim3(1), dim3(256), 0, stream, tab, (const float*)sumsq, T, max_norm, out);
  }
  hipLaunchKernelGGL(mt_scale_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, (const float*)out, (const float*)(sumsq + T),
                     sumsq_after + T);
  hipLaunchKernelGGL(mt_sumsq_finish_kernel, dim3(T), dim3(64), 0, stream, tab, chunk, nchunks, (const float*)(sumsq_after + T),
                     sumsq_after);
  return hmmc_launch_status();
}

extern "C" int hmmc_mt_bertadam(const long* tab, const float* groups_host, int ngroups, const int* chunk, int nchunks,
                                float* sumsq, int T, hipStream_t stream) {
  if (!tab || !groups_host || !chunk || !sumsq || nchunks <= 0 || T <= 0) return HMMC_ERR_ARG;
  if (ngroups <= 0 || ngroups > MAX_GROUPS) return HMMC_ERR_UNSUPPORTED;
  AdamGroups groups;
  for (int g = 0; g < ngroups; ++g)
    for (int j = 0; j < 8; ++j) groups.v[g][j] = groups_host[g * 8 + j];
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, sumsq + T);
  hipLaunchKernelGGL(mt_sumsq_finish_kernel, dim3(T), dim3(64), 0, stream, tab, chunk, nchunks, (const float*)(sumsq + T), sumsq);
  hipLaunchKernelGGL(mt_bertadam_kernel, dim3(nchunks), dim3(256), 0, stream, tab, groups, chunk, (const float*)sumsq, (const int*)nullptr);
  return hmmc_launch_status();
}

// BertAdam.step with the gradients' squared norms supplied: tensor t's is norms[index[t]] (norms = the sumsq_after of the
// hmmc_mt_clip_grad_norm_keep call that ran on the SAME, since unmodified, gradients; index maps this table's tensors to that
// call's).  Saves the optimizer's own pass over every gradient.
extern "C" int hmmc_mt_bertadam_ext(const long* tab, const float* groups_host, int ngroups, const int* chunk, int nchunks, int T,
                                    const float* norms, const int* index, hipStream_t stream) {
  if (!tab || !groups_host || !chunk || !norms || !index || nchunks <= 0 || T <= 0) return HMMC_ERR_ARG;
  if (ngroups <= 0 || ngroups > MAX_GROUPS) return HMMC_ERR_UNSUPPORTED;
  AdamGroups groups;
  for (int g = 0; g < ngroups; ++g)
    for (int j = 0; j < 8; ++j) groups.v[g][j] = groups_host[g * 8 + j];
  hipLaunchKernelGGL(mt_bertadam_kernel, dim3(nchunks), dim3(256), 0, stream, tab, groups, chunk, norms, index);
  return hmmc_launch_status();
}

extern "C" int hmmc_mt_ema(const long* tab, const int* chunk, int nchunks, float momentum, float one_minus_momentum,
                           hipStream_t stream) {
  if (!tab || !chunk || nchunks <= 0) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(mt_ema_kernel, dim3(nchunks), dim3(256), 0, stream, tab, chunk, momentum, one_minus_momentum);
  return hmmc_launch_status();
}

extern "C" int hmmc_mt_chunk_elems(void) { return CHUNK; }

extern "C" int hmmc_enqueue(const float* keys, float* queue, int R, int E, long W, long col0, hipStream_t stream) {
  if (!keys || !queue || R <= 0 || E <= 0 || col0 < 0 || col0 + R > W) return HMMC_ERR_ARG;
  hipLaunchKernelGGL(enqueue_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, keys, queue, R, E, W, col0);
  return hmmc_launch_status();
}
